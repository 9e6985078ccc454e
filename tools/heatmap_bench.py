#!/usr/bin/env python3
"""Times cogaim_heatmap_postprocess (B=32) against the numpy path it replaces.  Tuning aid only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd.inference import heatmap_postprocess  # noqa: E402

for N in (256, 1369, 5476):
    attn = torch.softmax(torch.randn(32, N, device="cuda"), dim=1)
    heatmap_postprocess(attn)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        heatmap_postprocess(attn)
    e1.record()
    torch.cuda.synchronize()
    a = attn.cpu().numpy()
    t0 = time.perf_counter()
    for b in range(32):
        c = np.power(a[b], 3)
        thr = np.percentile(c, 70)
        c = np.where(c > thr, c, c * 0.3)
        c = (c - c.min()) / (c.max() - c.min() + 1e-8)
    t1 = time.perf_counter()
    print(f"N={N}: device {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 32 images (incl. launch), numpy {(t1 - t0) * 1e6:.0f} us")
