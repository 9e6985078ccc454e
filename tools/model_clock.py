#!/usr/bin/env python3
"""Shader clock and duration of the persistent GEMM launches INSIDE the model's forward (library built with
-DCOGAIM_STAMPS, selected by COGAIM_LIB): two seconds of back-to-back forwards at the bench shape, then the stamps of
the last launch of each problem class.  Tuning / measurement aid only."""
import ctypes, json, os, statistics, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import _lib, create_model
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.cogaim_debug_clock.argtypes = [ctypes.c_void_p]
MODULES = ["ambient_stream", "iterative_focal_stream", "exif_prior_database"]
B = 32
x = torch.randn(B, 3, 518, 518, device="cuda")
exif = {"camera_idx": torch.zeros(B, dtype=torch.long, device="cuda"), "focal_length": torch.full((B,), 50.0, device="cuda"),
        "aperture": torch.full((B,), 2.8, device="cuda"), "iso": torch.full((B,), 100.0, device="cuda")}
out = {}
for dtype in ("bf16", "fp16"):
    for streams in (1, 2):
        m = create_model({"cognitive_modules": MODULES, "compute_dtype": dtype, "streams": streams}, {"num_cameras": 71}, device="cuda")
        for _ in range(3):
            m(x, exif)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 2.0
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() < t_end:
            for _ in range(10):
                m(x, exif)
            torch.cuda.synchronize(); n += 10
        ms = (time.perf_counter() - t0) / n * 1e3
        buf = np.zeros(4 * 256 * 4, dtype=np.int64)
        assert lib.cogaim_debug_clock(buf.ctypes.data) == 0
        res = {"ms_per_step": round(ms, 3)}
        for ci, name in enumerate(("fc1", "qkv", "fc2", "other")):
            b = buf.reshape(4, 256, 4)[ci]
            b = b[b[:, 3] > b[:, 1]]
            if len(b):
                ghz = statistics.median(float((r[2] - r[0]) / ((r[3] - r[1]) * 10.0)) for r in b)
                dur = (b[:, 3].max() - b[:, 1].min()) / 100.0
                res[name] = {"clock_ghz": round(ghz, 3), "first_entry_to_last_exit_us": round(float(dur), 1)}
        out[f"{dtype} streams={streams}"] = res
        print(dtype, streams, res, flush=True)
        del m
        torch.cuda.empty_cache()
print(json.dumps(out))
