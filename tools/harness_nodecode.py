#!/usr/bin/env python3
"""Upper bound of the batch loop without decode work: _decode returns arrays decoded beforehand.  Tuning aid only."""
import os, sys, tempfile, time
import numpy as np, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import inference as inf
from PIL import Image
tmp = tempfile.mkdtemp(prefix="cogaim_nd_")
rng = np.random.default_rng(0)
base = rng.integers(0, 256, size=(66, 66, 3), dtype=np.uint8)
big = np.asarray(Image.fromarray(base).resize((518, 518), Image.BILINEAR))
n = 1024
paths = []
for i in range(n):
    p = os.path.join(tmp, f"i{i:05d}.jpg")
    if i < 16:
        Image.fromarray(np.roll(big, 17 * i, axis=1)).save(p, quality=90)
    else:
        os.link(os.path.join(tmp, f"i{i % 16:05d}.jpg"), p)
    paths.append(p)
cfg_path = os.path.join(tmp, "cfg.yaml")
open(cfg_path, "w").write(yaml.safe_dump({"cognitive_modules": inf.DEFAULT_MODULES, "dataset": {"image_size": 518}, "compute_dtype": "bf16"}))
hn = inf.CognitiveAimInference(cfg_path, os.path.join(tmp, "none.pth"), device="cuda", batch_size=32, render=False)
hn.predict_batch(paths[:64])
cache = {p: hn._decode(p) for p in paths}
hn._decode = lambda p: cache[p]
torch.cuda.synchronize()
t0 = time.perf_counter()
hn.predict_batch(paths)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
hn.predict_batch(paths)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
print("no-decode harness img/s", round(n / dt, 1), {k: round(v, 3) for k, v in hn.last_timing.items()})
