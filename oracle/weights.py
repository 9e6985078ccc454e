"""Deterministic synthetic weights for parity tests.  TEST INFRASTRUCTURE ONLY.

The reference's pretrained weights are download links (README.md:9-12,
src/model.py:814) and cannot be obtained offline, so parity runs on seeded
synthetic weights.  Full-width (768) weights are too large to commit as
fixtures; instead every tensor is regenerated from (seed, key) here, both by
``oracle/make_goldens.py`` (which loads them into the real reference model)
and by the tests.

``reference_state_spec`` restates the reference's ``state_dict`` layout
(src/model.py:798-958 + the DINOv2 parameter names); ``make_goldens.py``
asserts it against the real module, which pins the key/shape contract.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch


def reference_state_spec(hidden: int = 768, layers: int = 12, patch: int = 14,
                         native_grid: int = 37, num_iterations: int = 3,
                         num_cameras: int = 71, mlp_ratio: int = 4,
                         modules=("ambient_stream", "iterative_focal_stream",
                                  "exif_prior_database")) -> "OrderedDict[str, Tuple[tuple, torch.dtype]]":
    D, f32 = hidden, torch.float32
    s: "OrderedDict[str, Tuple[tuple, torch.dtype]]" = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = ((o, i), f32)
        s[name + ".bias"] = ((o,), f32)

    e = "backbone.embeddings."
    s[e + "cls_token"] = ((1, 1, D), f32)
    s[e + "mask_token"] = ((1, D), f32)
    s[e + "position_embeddings"] = ((1, native_grid * native_grid + 1, D), f32)
    s[e + "patch_embeddings.projection.weight"] = ((D, 3, patch, patch), f32)
    s[e + "patch_embeddings.projection.bias"] = ((D,), f32)
    for i in range(layers):
        p = f"backbone.encoder.layer.{i}."
        s[p + "norm1.weight"] = ((D,), f32)
        s[p + "norm1.bias"] = ((D,), f32)
        for n in ("query", "key", "value"):
            lin(p + "attention.attention." + n, D, D)
        lin(p + "attention.output.dense", D, D)
        s[p + "layer_scale1.lambda1"] = ((D,), f32)
        s[p + "norm2.weight"] = ((D,), f32)
        s[p + "norm2.bias"] = ((D,), f32)
        lin(p + "mlp.fc1", D * mlp_ratio, D)
        lin(p + "mlp.fc2", D, D * mlp_ratio)
        s[p + "layer_scale2.lambda1"] = ((D,), f32)
    s["backbone.layernorm.weight"] = ((D,), f32)
    s["backbone.layernorm.bias"] = ((D,), f32)
    if "ambient_stream" in modules:
        lin("ambient_stream.mlp.0", 256, D)
        lin("ambient_stream.mlp.3", 128, 256)
        lin("ambient_stream.mlp.5", 64, 128)
    if "focal_stream" in modules and "iterative_focal_stream" not in modules:
        p = "focal_stream."
        s[p + "adaptive_weight"] = ((), f32)
        lin(p + "query_proj", D, D)
        lin(p + "key_proj", D, D)
        lin(p + "value_proj", D, D)
        lin(p + "curiosity_modulator.0", 32, 1)
        lin(p + "curiosity_modulator.2", 8, 32)
        lin(p + "projection.0", 256, D)
        lin(p + "projection.3", 64, 256)
    if "iterative_focal_stream" in modules:
        s["focal_stream.initial_focus"] = ((1, D), f32)
        for i in range(num_iterations):
            p = f"focal_stream.focal_streams.{i}."
            s[p + "adaptive_weight"] = ((), f32)
            lin(p + "query_proj", D, D)
            lin(p + "key_proj", D, D)
            lin(p + "value_proj", D, D)
            lin(p + "projection.0", 256, D)
            lin(p + "projection.3", 64, 256)
        lin("focal_stream.fusion.0", 128, 64 * num_iterations)
        lin("focal_stream.fusion.2", 64, 128)
    if "exif_prior_database" in modules and num_cameras:
        s["exif_prior.camera_embedding.weight"] = ((num_cameras, 64), f32)
        lin("exif_prior.exif_encoder.0", 64, 3)
        lin("exif_prior.exif_encoder.2", 64, 64)
        lin("exif_prior.fusion.0", 256, 128)
        lin("exif_prior.fusion.3", 64, 256)
    lin("fusion.0", 192, 192)
    for n in ("ambient", "focal", "exif"):
        lin(n + "_dim_aligner.projection", 768, 64)
    lin("decision_head.0", 1, 192)
    lin("confidence_head.0", 1, 192)
    lin("confidence_head.2", 1, 1)
    c = "curiosity_module."
    s[c + "curiosity_weights"] = ((3,), f32)
    s[c + "exploration_history"] = ((1000,), f32)
    s[c + "history_pointer"] = ((), torch.int64)
    lin(c + "encoder_mean.0", 384, 768)
    lin(c + "encoder_mean.3", 192, 384)
    lin(c + "encoder_logvar.0", 384, 768)
    lin(c + "encoder_logvar.3", 192, 384)
    lin(c + "decoder.0", 384, 192)
    lin(c + "decoder.3", 192, 384)
    lin(c + "uncertainty_head.0", 192, 768)
    lin(c + "uncertainty_head.2", 1, 192)
    lin(c + "geometric_curiosity.0", 256, 772)
    lin(c + "geometric_curiosity.2", 1, 256)
    lin(c + "local_curiosity.0", 128, 768)
    lin(c + "local_curiosity.2", 1, 128)
    lin("global_aligner.projection", 768, 2304)
    return s


def _randn(key: str, shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


def synth_tensor(key: str, shape, dtype, seed: int) -> torch.Tensor:
    """One tensor as a pure function of (seed, key).  Scales are chosen so
    activations stay O(1), softmaxes are neither uniform nor one-hot, and the
    confidence head's ReLU (src/model.py:938) stays active."""
    if dtype == torch.int64:
        return torch.zeros(tuple(shape), dtype=torch.int64)
    r = _randn(key, shape, seed)
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("exploration_history"):
        return torch.zeros(tuple(shape))
    if key.endswith("adaptive_weight"):
        return torch.tensor(0.5)
    if key.endswith("curiosity_weights"):
        return torch.tensor([0.4, 0.3, 0.3])
    if "norm" in key and leaf == "weight":
        return 1.0 + 0.1 * r
    if "norm" in key and leaf == "bias":
        return 0.1 * r
    if leaf == "lambda1":
        return 1.0 + 0.1 * r
    if key.endswith(("cls_token", "mask_token", "position_embeddings", "initial_focus")):
        return 0.02 * r if not key.endswith("position_embeddings") else 0.2 * r
    if key.endswith("patch_embeddings.projection.weight"):
        return r / math.sqrt(shape[1] * shape[2] * shape[3])
    if key.endswith("camera_embedding.weight"):
        return r
    if key == "confidence_head.0.bias":
        return torch.tensor([0.6])
    if key == "confidence_head.2.weight":
        return torch.tensor([[0.7]])
    if key == "confidence_head.2.bias":
        return torch.tensor([0.3])
    if key == "decision_head.0.bias":
        return torch.tensor([1.0])
    if key.startswith("exif_prior.exif_encoder.0.weight"):
        return 0.02 * r          # raw EXIF values are O(100), src/model.py:510-514
    if leaf == "bias":
        return 0.05 * r
    if leaf == "weight" and len(shape) == 2:
        if ".query_proj." in key or ".key_proj." in key or key.startswith(("focal_stream.query_proj", "focal_stream.key_proj")):
            return r * (1.2 * math.sqrt(2.0 / (shape[0] + shape[1])))   # src/model.py:379-380
        return r / math.sqrt(shape[1])
    return r


def synth_state_dict(spec: Dict[str, tuple], seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, synth_tensor(k, shp, dt, seed)) for k, (shp, dt) in spec.items())
