#!/usr/bin/env python3
"""Tools build only (tools/wt_build.sh): the 128x256 streaming GEMM kernel (cogaim_set_gemm_algo(3)) against the
256x256 persistent kernel (2) on the same operands -- outputs must be equal bit for bit -- and against an fp32
product on a sample of rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
WSB = torch.empty(call("cogaim_gemm_workspace_bytes"), device=dev, dtype=torch.uint8)
ok = True
for dt, tdt in ((0, torch.bfloat16), (1, torch.float16)):
    for kind, M, N, K in (("bias", 21920, 2304, 768), ("gelu", 21920, 3072, 768), ("gelu", 43840, 3072, 768),
                          ("bias", 43840, 2304, 768), ("bias", 43808, 768, 768), ("gelu", 16440, 1536, 1024),
                          ("bias", 12345, 1000, 3072), ("gelu", 33000, 3072, 832)):
        A = (torch.randn(M, K, device=dev) * 0.5).to(tdt)
        W = (torch.randn(N, K, device=dev) * 0.5).to(tdt)
        bias = torch.randn(N, device=dev)
        outs = []
        for algo in (2, 3):
            call("cogaim_set_gemm_algo", algo)
            out = torch.full((M, N), float("nan"), device=dev, dtype=tdt)
            for _ in range(2):       # the second run overwrites the first: a stale read of a parked tile would show
                if kind == "bias":
                    call("cogaim_gemm_bias_half", ptr(A), ptr(W), ptr(bias), ptr(out), M, N, K, K, K, N, (N // 12) * 4, 0.18, dt,
                         WSB.data_ptr(), WSB.numel(), None, stream_ptr())
                else:
                    call("cogaim_gemm_bias_gelu_half", ptr(A), ptr(W), ptr(bias), ptr(out), M, N, K, K, K, N, dt,
                         WSB.data_ptr(), WSB.numel(), None, stream_ptr())
            torch.cuda.synchronize()
            outs.append(out)
        rows = torch.tensor([0, 1, 63, 64, 127, 128, 255, 256, M // 2, M - 129, M - 65, M - 2, M - 1], device=dev)
        ref = A[rows].float() @ W.float().t() + bias
        if kind == "bias":
            ref[:, : (N // 12) * 4] *= 0.18
        else:
            ref = torch.nn.functional.gelu(ref)
        err = float((outs[1][rows].float() - ref).abs().max() / ref.abs().max())
        same = bool(torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)))
        nan = int(torch.isnan(outs[1].float()).sum())
        diff = int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum())
        print(f"dtype {dt} {kind} M={M} N={N} K={K}: bit-identical {same} (differing {diff}), nan {nan}, max-rel-err vs fp32 {err:.2e}", flush=True)
        ok = ok and same and nan == 0 and err < 2e-2
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
