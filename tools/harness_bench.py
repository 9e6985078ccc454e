#!/usr/bin/env python3
"""Images/s of the demo harness (files on disk -> decode -> upload -> device preprocessing -> batched forward ->
per-image results) next to the model-only rate on resident tensors.  Synthetic JPEGs, random-init weights.

    python tools/harness_bench.py [--n 512] [--src 518|4032x3024]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import inference as inf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--src", default="518")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from PIL import Image
    w, h = (int(v) for v in args.src.split("x")) if "x" in args.src else (int(args.src), int(args.src))
    tmp = tempfile.mkdtemp(prefix="cogaim_harness_")
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, size=(h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
    big = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))      # photo-like smoothness: realistic JPEG size
    n_unique = min(args.n, 16)
    for i in range(n_unique):
        Image.fromarray(np.roll(big, 17 * i, axis=1)).save(os.path.join(tmp, f"u{i}.jpg"), quality=90)
    paths = [os.path.join(tmp, f"u{i % n_unique}.jpg") for i in range(args.n)]
    size_kb = os.path.getsize(paths[0]) / 1024
    cfg = {"cognitive_modules": inf.DEFAULT_MODULES, "dataset": {"image_size": 518}, "compute_dtype": "bf16"}
    cfg_path = os.path.join(tmp, "cfg.yaml")
    open(cfg_path, "w").write(yaml.safe_dump(cfg))
    hn = inf.CognitiveAimInference(cfg_path, os.path.join(tmp, "no_checkpoint.pth"), device="cuda", batch_size=args.batch,
                                   render=False)
    # the harness treats a repeated path as one file; give every request its own name through hard links
    links = []
    for i, p in enumerate(paths):
        q = os.path.join(tmp, f"img{i:05d}.jpg")
        os.link(p, q)
        links.append(q)
    hn.predict_batch(links[: 2 * args.batch])                      # graph capture, tables
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = hn.predict_batch(links)
    torch.cuda.synchronize()
    t_h = time.perf_counter() - t0
    timing = {k: round(v, 4) for k, v in hn.last_timing.items()}
    assert all(r[0] is not None for r in res)
    x = torch.randn(args.batch, 3, 518, 518, device="cuda")
    exif = {k: v.cuda() for k, v in inf.exif_rows([None] * args.batch, hn.camera_to_id).items()}
    for _ in range(3):
        hn.model(x, exif)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        hn.model(x, exif)
    torch.cuda.synchronize()
    t_m = (time.perf_counter() - t0) / 10
    # decode rate alone (host threads)
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
        list(pool.map(hn._decode, links[: min(256, args.n)]))
    t_d = (time.perf_counter() - t0) / min(256, args.n)
    print(json.dumps({"source": f"{w}x{h} JPEG q90 ({size_kb:.0f} KB)", "images": args.n, "batch": args.batch,
                      "harness_img_s": round(args.n / t_h, 1), "model_only_img_s": round(args.batch / t_m, 1),
                      "decode_only_img_s": round(1.0 / t_d, 1), "decode_threads": min(16, os.cpu_count() or 4), "harness_wall_s": round(t_h, 4),
                      "harness_timing": timing}))


if __name__ == "__main__":
    main()
