#!/usr/bin/env python3
"""GPU micro-benchmark: attention output projection + LayerNorm as two launches (cogaim_gemm_residual_f32,
cogaim_layernorm) against the one-launch cogaim_gemm_residual_layernorm, at the bench shapes.  Tuning aid only."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr
torch.manual_seed(0)
ws = torch.empty(call("cogaim_gemm_workspace_bytes"), device="cuda", dtype=torch.uint8)
N = K = 768
for M in (21920, 43840, 5480):
    for di, tdt in ((0, torch.bfloat16), (1, torch.float16)):
        A = torch.randn(M, K, device="cuda").to(tdt)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(tdt)
        bias, ls = torch.randn(N, device="cuda"), 1 + 0.1 * torch.randn(N, device="cuda")
        g, b = 1 + 0.1 * torch.randn(N, device="cuda"), 0.1 * torch.randn(N, device="cuda")
        x = torch.randn(M, N, device="cuda")
        h = torch.empty(M, N, device="cuda", dtype=tdt)

        def two():
            call("cogaim_gemm_residual_f32", ptr(A), ptr(W), ptr(bias), ptr(ls), ptr(x), M, N, K, K, K, N, di, ws.data_ptr(),
                 ws.numel(), stream_ptr())
            call("cogaim_layernorm", ptr(x), ptr(g), ptr(b), ptr(h), M, N, 1e-6, di, stream_ptr())

        def one():
            call("cogaim_gemm_residual_layernorm", ptr(A), ptr(W), ptr(bias), ptr(ls), ptr(x), ptr(g), ptr(b), ptr(h), M, N,
                 K, K, K, N, N, 1e-6, di, stream_ptr())

        res = {}
        for name, fn in (("two launches", two), ("one launch", one)):
            x.normal_()
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                x.normal_()            # the residual stream grows with every in-place call otherwise
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            res[name] = statistics.median(ts) * 1000
        print(f"M={M} {('bf16', 'fp16')[di]}: two launches {res['two launches']:.1f} us, one launch {res['one launch']:.1f} us", flush=True)
