#!/usr/bin/env python3
"""Does an HBM-bound launch of one stream run UNDER a persistent GEMM of another stream, or behind it?  The GEMM holds one
512-thread workgroup on every CU; whether a LayerNorm wave fits beside its two waves per SIMD is a matter of registers
(512 per SIMD lane: 2 x 232 leave 48, 2 x 256 leave none).  Times, per pair: the GEMM alone (n launches back to back),
the LayerNorm alone (n launches), and both on two streams; 'hidden' is the share of the shorter one that disappeared."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
M, D = 21920, 768
dt, tdt = 0, torch.bfloat16
ws = torch.empty(call("cogaim_gemm_workspace_bytes"), device=dev, dtype=torch.uint8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def mk(m, k):
    return (torch.randn(m, k, device=dev) * 0.5).to(tdt)


A768, A3072 = mk(M, 768), mk(M, 3072)
W = {"qkv": mk(2304, 768), "fc1": mk(3072, 768), "out": mk(768, 768), "fc2": mk(768, 3072)}
bias = {k: torch.randn(v.shape[0], device=dev) for k, v in W.items()}
ls = torch.ones(768, device=dev)
out16 = torch.empty(M, 3072, device=dev, dtype=tdt)
x = torch.zeros(M, 768, device=dev)
x2 = torch.randn(M, 768, device=dev)
h = torch.empty(M, 768, device=dev, dtype=tdt)
g, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)


def gemm(kind, st):
    sp = st.cuda_stream
    if kind == "qkv":
        call("cogaim_gemm_bias_half", ptr(A768), ptr(W[kind]), ptr(bias[kind]), ptr(out16), M, 2304, 768, 768, 768, 2304, 768, 0.18, dt,
             ws.data_ptr(), ws.numel(), None, sp)
    elif kind == "fc1":
        call("cogaim_gemm_bias_gelu_half", ptr(A768), ptr(W[kind]), ptr(bias[kind]), ptr(out16), M, 3072, 768, 768, 768, 3072, dt,
             ws.data_ptr(), ws.numel(), None, sp)
    elif kind == "out":
        call("cogaim_gemm_residual_f32", ptr(A768), ptr(W[kind]), ptr(bias[kind]), ptr(ls), ptr(x), M, 768, 768, 768, 768, 768, dt,
             ws.data_ptr(), ws.numel(), sp)
    else:
        call("cogaim_gemm_residual_f32", ptr(A3072), ptr(W[kind]), ptr(bias[kind]), ptr(ls), ptr(x), M, 768, 3072, 3072, 3072, 768, dt,
             ws.data_ptr(), ws.numel(), sp)


def ln(st):
    call("cogaim_layernorm", ptr(x2), ptr(g), ptr(b), ptr(h), M, D, 1e-6, dt, st.cuda_stream)


def timed(fn_a, fn_b, n_a, n_b):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        s1.wait_event(e0); s2.wait_event(e0)
        for _ in range(n_a):
            fn_a()
        for _ in range(n_b):
            fn_b()
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(s1); eb.record(s2)
        torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000)
    return best


for kind in ("qkv", "fc1", "out", "fc2"):
    n = 20
    ta = timed(lambda: gemm(kind, s1), lambda: None, n, 0)
    nl = max(1, int(ta / 16.0))                          # about as much LayerNorm time as GEMM time
    tb = timed(lambda: None, lambda: ln(s2), 0, nl)
    tab = timed(lambda: gemm(kind, s1), lambda: ln(s2), n, nl)
    hidden = (ta + tb - tab) / min(ta, tb)
    print(f"{kind}: {n} GEMMs {ta:.0f} us, {nl} LayerNorms {tb:.0f} us, both {tab:.0f} us, hidden {100 * hidden:.0f} %", flush=True)
