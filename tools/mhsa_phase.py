#!/usr/bin/env python3
"""Reads the per-phase cycle totals an MHSA_ABLATE=9 build leaves in the first rows of ctx.  Tuning aid only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr
B, T, H = 32, 1370, 12
D = 64 * H
torch.manual_seed(0)
for dt, tdt in ((0, torch.bfloat16), (1, torch.float16)):
    ctx = torch.zeros(B * T, D, device="cuda", dtype=tdt)
    qkv = (torch.randn(B * T, 3 * D, device="cuda") * 0.6).to(tdt)
    for _ in range(3):
        call("cogaim_mhsa_qkv_fwd", ptr(qkv), ptr(ctx), B, T, H, 3 * D, dt, stream_ptr())
    torch.cuda.synchronize()
    nblk = 11 * B * H
    n = (nblk - 6) // 97 + 1
    raw = ctx.view(-1)[: n * 4 * 8 * 4].view(torch.int64).view(n, 4, 8)[:, :, :6].cpu().double()
    per_tile = raw / 22.0
    names = ["Kfrag reads", "QK mfma", "mask+max", "rescale+exp+PV", "store tile", "barrier"]
    m = per_tile.mean(dim=(0, 1))
    print(f"dtype {dt}: cycles per key tile per wave (mean over {n} blocks x 4 waves), total {m.sum():.0f}")
    for nm, v in zip(names, m.tolist()):
        print(f"   {nm:18s} {v:8.1f}")
