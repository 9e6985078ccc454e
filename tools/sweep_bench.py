#!/usr/bin/env python3
"""Latency of the 9-instruction sweep over one image (one backbone + focal pass, nine masks) against nine single
guided calls, at 224 and 518 pixels, and for a batch of 8 images.  Writes one JSON line.  Measurement aid only."""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import create_model
MODULES = ["ambient_stream", "iterative_focal_stream", "exif_prior_database"]
NAMES = ["center", "left", "right", "top", "bottom", "top-left", "top-right", "bottom-left", "bottom-right"]
model = create_model({"cognitive_modules": MODULES, "compute_dtype": "bf16"}, {"num_cameras": 71}, device="cuda")
torch.manual_seed(0)
proj = (torch.randn(64, 768) * 0.02, torch.zeros(64))
out = {}
for B, S in ((1, 224), (1, 518), (8, 518)):
    x = torch.randn(B, 3, S, S, device="cuda")
    exif = {"camera_idx": torch.zeros(B, dtype=torch.long, device="cuda"), "focal_length": torch.full((B,), 50.0, device="cuda"),
            "aperture": torch.full((B,), 2.8, device="cuda"), "iso": torch.full((B,), 100.0, device="cuda")}
    def sweep():
        return model.forward_with_guidance(x, exif, NAMES, return_attention=True, temp_projection=proj)
    def singles():
        return [model.forward_with_guidance(x, exif, n, return_attention=True, temp_projection=proj) for n in NAMES]
    res = {}
    for name, f in (("sweep", sweep), ("nine_calls", singles)):
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            r = f()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[name + "_ms"] = round(statistics.median(ts), 3)
    D, C, A = sweep()
    one = singles()
    torch.cuda.synchronize()
    res["identical"] = bool(all(torch.equal(A[k], one[k][2]) and torch.equal(D[k], one[k][0]) for k in range(9)))
    res["speedup"] = round(res["nine_calls_ms"] / res["sweep_ms"], 2)
    out[f"B={B} {S}x{S}"] = res
    print(f"B={B} {S}x{S}", res, flush=True)
print(json.dumps(out))
