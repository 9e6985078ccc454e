#!/usr/bin/env python3
"""Runs N complete forwards of the bench configuration (after 3 that warm up and capture) so that two rocprofv3
kernel traces with different N show which kernels a single steady-state forward launches, copies included:
    rocprofv3 --kernel-trace --stats -d gpurun_out/k10 -- python3 tools/count_kernels.py 10
Tuning aid only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import create_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B = 32
images = torch.randn(B, 3, 518, 518, device=dev)
exif = {"camera_idx": torch.zeros(B, dtype=torch.long, device=dev), "focal_length": torch.full((B,), 50.0, device=dev),
        "aperture": torch.full((B,), 2.8, device=dev), "iso": torch.full((B,), 100.0, device=dev)}
cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"], "compute_dtype": "bf16",
       "focal_config": {"num_iterations": 3, "focus_strength": 1.5}, "init_seed": 0}
model = create_model(cfg, {"num_cameras": 71}, device=dev)
for _ in range(3 + n):
    model(images, exif)
torch.cuda.synchronize()
