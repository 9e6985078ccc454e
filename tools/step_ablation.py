#!/usr/bin/env python3
"""Timing-only ablation of the forward step (GPU, tuning aid; results of the ablated runs are WRONG by design).

Which launches is the step's wall time made of once two sub-batch streams overlap everything?  For each variant a set
of C-ABI launches is dropped (their outputs keep the realistic values a complete forward left in the workspace), the
launch sequence is re-captured and 20 steps are timed exactly as bench.py times them.  The difference to the complete
step is an UPPER bound on what fusing, hiding or speeding up that group could gain.

    python tools/step_ablation.py [--batch 32] [--size 518] [--dtype bf16] [--steps 20] > gpurun_out/ablation.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import _lib, create_model  # noqa: E402
from cognitive_aim_depth_estimation_amd import model as model_mod  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=518)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--only", default="")
args = ap.parse_args()

D = 768
SMALL = {"cogaim_mlp_chain", "cogaim_linear_f32", "cogaim_fusion_heads"}
FOCAL = {"cogaim_focal_prep", "cogaim_gemm_batched_f32", "cogaim_focal_softmax_stats_colsum", "cogaim_focal_attention_candidates",
         "cogaim_focal_attention_vector", "cogaim_focal_weighted_colsum", "cogaim_focal_reduce_u", "cogaim_weighted_tokens", "cogaim_weighted_tokens_parts"}


def is_focal_proj(name, a):
    return name == "cogaim_gemm_bias_half" and a[5] == D


VARIANTS = {
    "complete": lambda n, a: False,
    "no_layernorm": lambda n, a: n == "cogaim_layernorm",
    "no_small_mlps": lambda n, a: n in SMALL,
    "no_focal_stream": lambda n, a: n in FOCAL or n in SMALL or is_focal_proj(n, a),
    "no_mhsa": lambda n, a: n == "cogaim_mhsa_qkv_fwd",
    "no_qkv_gemm": lambda n, a: n == "cogaim_gemm_bias_half" and a[5] == 3 * D,
    "no_fc1_gemm": lambda n, a: n == "cogaim_gemm_bias_gelu_half",
    "no_out_proj_gemm": lambda n, a: n == "cogaim_gemm_residual_f32" and a[7] == D,
    "no_fc2_gemm": lambda n, a: n == "cogaim_gemm_residual_f32" and a[7] == 4 * D,
    "no_patch_embed": lambda n, a: n in ("cogaim_im2col", "cogaim_gemm_patch_embed"),
    "only_gemms_and_mhsa": lambda n, a: not (n.startswith("cogaim_gemm_") and not is_focal_proj(n, a) and n != "cogaim_gemm_batched_f32"
                                              and n != "cogaim_gemm_patch_embed" or n == "cogaim_mhsa_qkv_fwd" or n == "cogaim_export_outputs"),
}
if args.only:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in args.only.split(",") or k == "complete"}

real_call = _lib.call
state = {"drop": VARIANTS["complete"]}


def filtered(name, *a):
    if state["drop"](name, a):
        return 0
    return real_call(name, *a)


_lib.call = filtered
model_mod.call = filtered

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B = args.batch
g = torch.Generator(device=dev).manual_seed(1234)
images = torch.randn(B, 3, args.size, args.size, device=dev, generator=g)
exif = {"camera_idx": torch.zeros(B, dtype=torch.long, device=dev), "focal_length": torch.full((B,), 50.0, device=dev),
        "aperture": torch.full((B,), 2.8, device=dev), "iso": torch.full((B,), 100.0, device=dev)}
cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"], "compute_dtype": args.dtype,
       "focal_config": {"num_iterations": 3, "focus_strength": 1.5}, "init_seed": 0, "streams": args.streams}
model = create_model(cfg, {"num_cameras": 71}, device=dev)
out = {"batch": B, "size": args.size, "dtype": args.dtype, "streams": args.streams, "steps": args.steps, "variants": {}}
for name, drop in VARIANTS.items():
    state["drop"] = VARIANTS["complete"]
    for ws in model._ws.values():
        ws.pop("graphs", None)
    model(images, exif)                       # complete forward: every buffer holds realistic values
    torch.cuda.synchronize()
    state["drop"] = drop
    for ws in model._ws.values():
        ws.pop("graphs", None)
    try:
        for _ in range(2 + 3):                # eager, capture, warm-up replays
            model(images, exif)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(images, exif)
        torch.cuda.synchronize()
        ms = 1000.0 * (time.perf_counter() - t0) / args.steps
    except Exception as e:                    # a deferred device-side condition raised by garbage values
        ms = None
        out["variants"][name] = {"error": repr(e)[:200]}
        continue
    finally:
        for ws in model._ws.values():         # drop pending flag snapshots of the ablated runs
            for slot in ws["flag_ring"]:
                slot["pending"] = False
            ws["flags"].zero_()
    out["variants"][name] = {"ms_per_step": round(ms, 3)}
base = out["variants"]["complete"]["ms_per_step"]
for k, v in out["variants"].items():
    if "ms_per_step" in v:
        v["saved_ms"] = round(base - v["ms_per_step"], 3)
        v["saved_pct"] = round(100.0 * (base - v["ms_per_step"]) / base, 2)
print(json.dumps(out, indent=1))
