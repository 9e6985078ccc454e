#!/usr/bin/env python3
"""GPU micro-benchmark of cogaim_im2col at the bench shapes.  Tuning aid only."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr
for B, S in ((16, 518), (32, 518), (4, 1036)):
    g = S // 14
    img = torch.randn(B, 3, S, S, device="cuda")
    cols = torch.empty(B * g * g, 640, device="cuda", dtype=torch.float16)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call("cogaim_im2col", ptr(img), ptr(cols), B, S, S, 14, 640, 1, stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    byt = img.numel() * 4 + cols.numel() * 2
    print(f"im2col B={B} {S}px: {ms*1000:.1f} us  {byt/ms/1e9:.2f} TB/s (pixels + cols)")
