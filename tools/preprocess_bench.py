#!/usr/bin/env python3
"""Times cogaim_preprocess_rgb8 (device resize + normalise) against the host path on one photo-sized image."""
import os, sys, time
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd.inference import DevicePreprocessor, preprocess_image
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr

for (h, w), S in (((3024, 4032), 518), ((1080, 1920), 518), ((480, 640), 518), ((480, 640), 224)):
    arr = np.random.default_rng(0).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    im = Image.fromarray(arr, "RGB")
    t0 = time.perf_counter(); ref = preprocess_image(im, (S, S)); t_host = time.perf_counter() - t0
    pre = DevicePreprocessor((S, S), "cuda")
    out = pre(im); torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)
    src = torch.from_numpy(arr).cuda()
    xb, xc, xk = pre._axis_tables(w, S); yb, yc, yk = pre._axis_tables(h, S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        call("cogaim_preprocess_rgb8", ptr(src), h, w, w * 3, 3, ptr(xb), ptr(xc), xk, ptr(yb), ptr(yc), yk,
             ptr(pre._tmp), ptr(pre._lut), ptr(out), S, S, stream_ptr())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    algo = h * w * 3 + 2 * h * S * 3 + S * S * 12
    print(f"{h}x{w} -> {S}: device kernels {us:.1f} us ({algo / us / 1e3:.0f} GB/s on {algo / 1e6:.1f} MB algorithmic), "
          f"host path {t_host * 1e3:.1f} ms")
