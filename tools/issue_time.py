#!/usr/bin/env python3
"""Host time to ISSUE one forward (call returns; the GPU still runs) vs the GPU time of the step.  Tuning aid only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import create_model
MODULES = ["ambient_stream", "iterative_focal_stream", "exif_prior_database"]
m = create_model({"cognitive_modules": MODULES, "compute_dtype": "bf16"}, {"num_cameras": 71}, device="cuda")
B = 32
x = torch.randn(B, 3, 518, 518, device="cuda")
exif = {"camera_idx": torch.zeros(B, dtype=torch.long, device="cuda"), "focal_length": torch.full((B,), 50.0, device="cuda"),
        "aperture": torch.full((B,), 2.8, device="cuda"), "iso": torch.full((B,), 100.0, device="cuda")}
for _ in range(4):
    m(x, exif)
torch.cuda.synchronize()
iss, tot = [], []
for _ in range(10):
    t0 = time.perf_counter()
    m(x, exif)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("issue ms", [round(v, 2) for v in iss])
print("issue+run ms", [round(v, 2) for v in tot])
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m(x, exif)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
