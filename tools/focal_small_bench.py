#!/usr/bin/env python3
"""GPU micro-benchmark of cogaim_focal_prep and cogaim_weighted_tokens at the bench shapes.  Tuning aid only."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr
torch.manual_seed(0)
B, N, D = 16, 1369, 768
T = N + 1
tok = torch.randn(B, T, D, device="cuda")
pe = torch.randn(N, D, device="cuda")
scale = 1 + torch.rand(B, N, device="cuda")
w = torch.rand(B, N, device="cuda")
out16 = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16)
wt = torch.empty(B, D, device="cuda")
rows = call("cogaim_weighted_tokens_rows")
part = torch.empty(B * ((N + rows - 1) // rows) * D, device="cuda")
def timeit(f):
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    return statistics.median(ts) * 1000
prep = lambda: call("cogaim_focal_prep", tok.data_ptr() + D * 4, T * D, ptr(scale), ptr(pe), ptr(out16), B, N, D, 0, stream_ptr())
wtk = lambda: call("cogaim_weighted_tokens", tok.data_ptr() + D * 4, T * D, ptr(scale), ptr(pe), ptr(w), ptr(wt), ptr(part), B, N, D, stream_ptr())
prep(); wtk(); torch.cuda.synchronize()
ref_prep = (tok[:, 1:] * scale[..., None] + pe).to(torch.bfloat16).reshape(B * N, D)
ref_wt = torch.einsum("bn,bnd->bd", w.double(), (tok[:, 1:] * scale[..., None] + pe).double())
print(f"focal_prep {timeit(prep):.1f} us  exact {bool(torch.equal(out16, ref_prep))};  weighted_tokens {timeit(wtk):.1f} us  "
      f"max rel err {float(((wt.double() - ref_wt).abs() / ref_wt.abs().clamp_min(1e-6)).max()):.2e}  checksum {float(wt.double().sum()):.10f}")
