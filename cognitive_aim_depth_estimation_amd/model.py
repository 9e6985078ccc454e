"""MI355X-native Cognitive-Aim model behind the reference's nn.Module API.

Drop-in for reference ``src/model.py``: ``create_model(config, camera_info,
device)`` (:1534), ``CognitiveAimModel.forward(images, exif_data,
return_attention)`` (:1064), ``forward_with_guidance`` (:1157),
``get_attention_weights`` (:1058), ``get_features`` (:1428), the
``use_ambient / use_focal / use_exif`` attributes and a ``state_dict`` with the
reference's exact keys and shapes, so ``demo.py``-style checkpoint loading
works unchanged.

All arithmetic runs in hand-written gfx950 kernels (libcogaim.so, bound in
``_lib.py``); torch only owns device memory and streams.  There is no CPU or
eager fallback: tensors must live on a ROCm device.

Differences from the reference that are deliberate:
  * each call runs the backbone and the iterative focal stream once; the
    reference repeats them 2-4x per call with bit-identical results
    (:1008/:1021/:1095/:1129);
  * the curiosity module (:521-793) is not evaluated: its score never reaches
    the outputs when ``curiosity_guided`` is off (the only setting the shipped
    configs produce) and cancels in the renormalisation of :282 otherwise; its
    parameters and buffers are kept for ``state_dict`` compatibility;
  * kernel launch errors raise instead of being turned into all-zero features
    (:1050-1056).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import tables
from ._lib import call, mlp_chain, ptr, stream_ptr

LOG2E = 1.4426950408889634
_DTYPES = {"bf16": (0, torch.bfloat16), "fp16": (1, torch.float16)}
# device-side condition words (int32 each) of a workspace: one per focal iteration, then the named ones
_FLAG_WORDS = 64
_MAX_ITERATIONS = 48
_FLAG_RANGE = 60        # a 16-bit store exceeded the fp16 range (fp16 operands only)
_FLAG_CAMERA = 63       # camera_idx outside the embedding table
_FLAG_SNAPSHOTS = 4     # host copies of the flag words in flight per workspace (bounds how far the host runs ahead)


# --------------------------------------------------------------------------
# parameter layout
# --------------------------------------------------------------------------
def state_spec(hidden: int, layers: int, patch: int, native_grid: int, num_iterations: int,
               num_cameras: int, use_ambient: bool, use_focal: bool, use_exif: bool,
               curiosity_guided: bool, iterative: bool = True) -> "OrderedDict[str, Tuple[tuple, torch.dtype]]":
    """Key -> (shape, dtype) in the reference's registration order."""
    D, f32 = hidden, torch.float32
    s: "OrderedDict[str, Tuple[tuple, torch.dtype]]" = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = ((o, i), f32)
        s[name + ".bias"] = ((o,), f32)

    e = "backbone.embeddings."
    s[e + "cls_token"] = ((1, 1, D), f32)
    s[e + "mask_token"] = ((1, D), f32)
    s[e + "position_embeddings"] = ((1, native_grid ** 2 + 1, D), f32)
    s[e + "patch_embeddings.projection.weight"] = ((D, 3, patch, patch), f32)
    s[e + "patch_embeddings.projection.bias"] = ((D,), f32)
    for i in range(layers):
        p = f"backbone.encoder.layer.{i}."
        for n in ("norm1.weight", "norm1.bias"):
            s[p + n] = ((D,), f32)
        for n in ("query", "key", "value"):
            lin(p + "attention.attention." + n, D, D)
        lin(p + "attention.output.dense", D, D)
        s[p + "layer_scale1.lambda1"] = ((D,), f32)
        for n in ("norm2.weight", "norm2.bias"):
            s[p + n] = ((D,), f32)
        lin(p + "mlp.fc1", 4 * D, D)
        lin(p + "mlp.fc2", D, 4 * D)
        s[p + "layer_scale2.lambda1"] = ((D,), f32)
    s["backbone.layernorm.weight"] = ((D,), f32)
    s["backbone.layernorm.bias"] = ((D,), f32)
    if use_ambient:
        lin("ambient_stream.mlp.0", 256, D)
        lin("ambient_stream.mlp.3", 128, 256)
        lin("ambient_stream.mlp.5", 64, 128)
    if use_focal and not iterative:
        # a bare FocalStream (src/model.py:869-870); its constructor default keeps the curiosity modulator (:58, :73-79)
        p = "focal_stream."
        s[p + "adaptive_weight"] = ((), f32)
        for n in ("query_proj", "key_proj", "value_proj"):
            lin(p + n, D, D)
        lin(p + "curiosity_modulator.0", 32, 1)
        lin(p + "curiosity_modulator.2", 8, 32)
        lin(p + "projection.0", 256, D)
        lin(p + "projection.3", 64, 256)
    elif use_focal:
        s["focal_stream.initial_focus"] = ((1, D), f32)
        for i in range(num_iterations):
            p = f"focal_stream.focal_streams.{i}."
            s[p + "adaptive_weight"] = ((), f32)
            for n in ("query_proj", "key_proj", "value_proj"):
                lin(p + n, D, D)
            if curiosity_guided:
                lin(p + "curiosity_modulator.0", 32, 1)
                lin(p + "curiosity_modulator.2", 8, 32)
            lin(p + "projection.0", 256, D)
            lin(p + "projection.3", 64, 256)
        if curiosity_guided:
            lin("focal_stream.curiosity_amplifier.0", 32, 1)
            lin("focal_stream.curiosity_amplifier.2", num_iterations, 32)
        lin("focal_stream.fusion.0", 128, 64 * num_iterations)
        lin("focal_stream.fusion.2", 64, 128)
    if use_exif:
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
    for n, o, i in (("encoder_mean.0", 384, 768), ("encoder_mean.3", 192, 384),
                    ("encoder_logvar.0", 384, 768), ("encoder_logvar.3", 192, 384),
                    ("decoder.0", 384, 192), ("decoder.3", 192, 384),
                    ("uncertainty_head.0", 192, 768), ("uncertainty_head.2", 1, 192),
                    ("geometric_curiosity.0", 256, 772), ("geometric_curiosity.2", 1, 256),
                    ("local_curiosity.0", 128, 768), ("local_curiosity.2", 1, 128)):
        lin(c + n, o, i)
    lin("global_aligner.projection", 768, 2304)
    return s


_BUFFERS = ("curiosity_module.exploration_history", "curiosity_module.history_pointer")


def _init_tensor(key: str, shape, dtype, gen: torch.Generator) -> torch.Tensor:
    """Random init following the reference's distributions (src/model.py:95-126,
    351-389, 931-945; DINOv2 trunc-normal(0.02) / LayerScale 1.0, hf:401-414).
    Pretrained weights cannot be fetched offline; real use loads a checkpoint."""
    leaf = key.rsplit(".", 1)[-1]
    if dtype == torch.int64:
        return torch.zeros(shape, dtype=dtype)
    if key.endswith("exploration_history"):
        return torch.zeros(shape)
    if key.endswith("adaptive_weight"):
        return torch.tensor(0.5)
    if key.endswith("curiosity_weights"):
        return torch.tensor([0.4, 0.3, 0.3])
    if key.startswith("backbone."):
        if "norm" in key:
            return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
        if leaf == "lambda1":
            return torch.ones(shape)
        if leaf == "bias" or leaf == "mask_token":
            return torch.zeros(shape)
        return torch.nn.init.trunc_normal_(torch.empty(shape), std=0.02, generator=gen)
    if key == "focal_stream.initial_focus":
        return torch.randn(shape, generator=gen) * 0.02
    if key == "decision_head.0.bias":
        return torch.ones(shape)
    if key == "confidence_head.2.bias":
        return torch.full(shape, 2.0)
    if leaf == "weight" and len(shape) == 2:
        o, i = shape
        if key.endswith("camera_embedding.weight"):
            return torch.randn(shape, generator=gen)
        if ".focal_streams." in key and ("query_proj" in key or "key_proj" in key or "value_proj" in key):
            it = int(key.split(".focal_streams.")[1].split(".")[0])
            gain = (1.0 if "value_proj" in key else 1.2) * (1.0 + 0.1 * it)
            return torch.randn(shape, generator=gen) * (gain * math.sqrt(2.0 / (i + o)))
        if key in ("focal_stream.query_proj.weight", "focal_stream.key_proj.weight", "focal_stream.value_proj.weight"):
            gain = 1.0 if "value_proj" in key else 2.0                     # src/model.py:116-118
            return torch.randn(shape, generator=gen) * (gain * math.sqrt(2.0 / (i + o)))
        if key.startswith("focal_stream.") or key == "decision_head.0.weight":
            gain = 1.0 if key == "decision_head.0.weight" else 0.8
            bound = gain * math.sqrt(6.0 / (i + o))
            return (torch.rand(shape, generator=gen) * 2 - 1) * bound
        bound = 1.0 / math.sqrt(i)       # nn.Linear default (kaiming_uniform a=sqrt(5))
        return (torch.rand(shape, generator=gen) * 2 - 1) * bound
    if leaf == "bias":
        if ".focal_streams." in key and ("query_proj" in key or "key_proj" in key):
            it = int(key.split(".focal_streams.")[1].split(".")[0])
            return (torch.rand(shape, generator=gen) * 2 - 1) * 0.01 * (1.0 + 0.1 * it)
        if key.startswith("focal_stream."):
            return torch.zeros(shape)
        return torch.zeros(shape)
    return torch.randn(shape, generator=gen) * 0.02


class _Pending:
    """Handle of CognitiveAimModel.forward_async."""

    def __init__(self, model, outs, done, stream, return_attention, ready=None):
        self._model, self._outs, self._done, self._stream = model, outs, done, stream
        self._return_attention, self._ready = return_attention, ready

    @property
    def done_event(self):
        """Event after the batch's last kernel (None when the call ran synchronously)."""
        return self._done

    def result(self):
        """(depth, conf[, attention]) as forward returns them, valid on the caller's current stream."""
        if self._ready is not None:
            return self._ready
        depth, conf, attn, fused = self._outs
        cur = torch.cuda.current_stream(depth.device)
        cur.wait_event(self._done)
        for t in (depth, conf, attn, fused):
            if t is not None:
                t.record_stream(cur)
        m = self._model
        m.fusion_features = fused
        if m.use_focal and not hasattr(m, "_last_attention_weights"):
            m._last_attention_weights = attn
        elif not m.use_focal:
            m._last_attention_weights = None
        self._ready = (depth, conf, attn) if self._return_attention else (depth, conf)
        return self._ready


class _Node(nn.Module):
    """Bare container: exists only to give parameters the reference's dotted names."""


# --------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------
class CognitiveAimModel(nn.Module):
    def __init__(self, config: dict, camera_info: Optional[dict] = None):
        super().__init__()
        self.config = config
        # -- config resolution, same keys and defaults as reference src/model.py:803-951 --
        self.backbone_size = config.get("backbone_size", "base")
        large = self.backbone_size == "large"          # anything else is dinov2-base (src/model.py:804-812)
        self.feature_dim = 1024 if large else 768
        self.use_lora = config.get("use_lora", False)
        if self.use_lora:
            raise NotImplementedError("LoRA is unused and broken in the reference (src/model.py:30)")
        model_cfg = config.get("model", {}) or {}
        modules = model_cfg.get("cognitive_modules", config.get("cognitive_modules", []))
        self.use_ambient = "ambient_stream" in modules
        cg = config.get("curiosity_guided_attention", {}) or {}
        self.curiosity_guided = bool(cg.get("enabled", False))
        focal_cfg = config.get("focal_config", {}) or {}
        if "iterative_focal_stream" in modules:
            self.use_focal, self.use_iterative = True, True
            self.num_iterations = int(focal_cfg.get("num_iterations", 3))
            self.focus_strength = float(focal_cfg.get("focus_strength", 1.5))
            if not 1 <= self.num_iterations <= _MAX_ITERATIONS:
                raise ValueError(f"focal_config.num_iterations must be in 1..{_MAX_ITERATIONS}")
            if config.get("focal_hidden_dim", 256) != 256:
                raise NotImplementedError("focal_hidden_dim != 256")
        elif "focal_stream" in modules:                       # one bare FocalStream (src/model.py:869-873)
            self.use_focal, self.use_iterative = True, False
            self.num_iterations, self.focus_strength = 1, 0.0
        else:
            self.use_focal, self.use_iterative = False, False
            self.num_iterations, self.focus_strength = 0, 0.0
        self.use_exif = bool("exif_prior_database" in modules and camera_info)
        num_cameras = int(camera_info["num_cameras"]) if self.use_exif else 0
        self.fusion_dim = 192
        self.target_fusion_dim = 768
        # -- backbone geometry: dinov2-base unless overridden (tests use shallow stacks) --
        bb = config.get("backbone_config", {}) or {}
        self.num_layers = int(bb.get("num_hidden_layers", 24 if large else 12))
        self.num_heads = int(bb.get("num_attention_heads", 16 if large else 12))
        self.patch_size = int(bb.get("patch_size", 14))
        self.native_grid = int(bb.get("image_size", 518)) // self.patch_size
        self.layer_norm_eps = float(bb.get("layer_norm_eps", 1e-6))
        if self.feature_dim != 64 * self.num_heads:
            raise NotImplementedError("attention kernel is built for head_dim 64")
        self.compute_dtype = str(config.get("compute_dtype", os.environ.get("COGAIM_DTYPE", "fp16")))
        if self.compute_dtype not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        # Batches of 2 x 8 images or more are run as `streams` sub-batches on streams of the model's own: the kernel
        # boundaries, partial last rounds and store-bound epilogues of one sub-batch's launches fill with the other's
        # work (same-box A/B at B = 32, 518 pixels: 2 streams +2 %: each half's GEMMs run 12 % slower than the whole
        # batch's, the LayerNorm / focal / tail time hides).  Per-image results do not depend on batch mates; they
        # are bit-identical to a single-stream run unless a GEMM K-splits its partial last round of tiles (K = 3072 at
        # production sizes: which rows lie in that round depends on the row count, and a split tile sums its K in
        # slices -- the same value to fp32 rounding).
        self.num_streams = max(1, int(config.get("streams", os.environ.get("COGAIM_STREAMS", "2"))))
        self._side_streams = None
        # Opt-in: evaluate the curiosity module wherever the reference does (src/model.py:995, :1104, :1139, :1185),
        # for the state it leaves behind -- exploration_history / history_pointer, get_exploration_stats() -- and for
        # the draws it takes from torch's CPU generator (:609, :744), which fix where the guided path's per-call
        # nn.Linear(768, 64) (:1417-1422) falls in the random stream.  Its score still reaches no output.
        # The module is 768 wide whatever the backbone (:946-950): on the 1024 columns of backbone_size='large' every
        # call raises before it draws or stores anything and the reference catches that (:997, :1141), so there the
        # faithful mode has nothing to do.
        self.faithful_curiosity = bool(config.get("faithful_curiosity", False)) and not large
        # largest patch grid whose fp32 focal scores are kept in memory (see _workspace); tests lower it
        self.scores_in_memory_max_patches = int(config.get("focal_scores_in_memory_max_patches", 1536))
        # Outward error behaviour.  The reference swallows every exception of its feature extraction and carries on
        # with all-zero fused features (src/model.py:1050-1056), and answers any exception of the guided forward with
        # the standard forward (:1237-1240).  Errors found on the HOST before anything is launched -- an EXIF dict
        # without one of its four keys or with the wrong number of entries, a guidance tensor that cannot be laid out
        # on the patch grid -- are treated the same way unless ``strict_errors`` is set; then they raise.  Kernel
        # launch errors and device-side conditions always raise.
        self.strict_errors = bool(config.get("strict_errors", False))
        # workspaces (one per batch shape and stream, each with its captured graphs) kept before the least recently
        # used one is dropped; ~1 GB each at 32 images of 518 x 518
        self.workspace_cache = max(1, int(config.get("workspace_cache", 12)))
        # attention output projection and the LayerNorm after it as one kernel (width 768 only).  Off by default: the
        # kernel owns whole rows, so every workgroup streams all of W through LDS, and inside the two-stream forward
        # at 16 x 518 x 518 it measured 15 us per layer slower than the two launches (alone it ties them at 68 us).
        self.fuse_out_proj_layernorm = bool(config.get("fuse_out_proj_layernorm", False))
        # replay the per-shape launch sequence as one HIP graph after its first (eager) run
        self.use_graph = bool(config.get("use_hip_graph", os.environ.get("COGAIM_GRAPH", "1") != "0"))

        spec = state_spec(self.feature_dim, self.num_layers, self.patch_size, self.native_grid,
                          self.num_iterations, num_cameras, self.use_ambient, self.use_focal,
                          self.use_exif, self.curiosity_guided, self.use_iterative or not self.use_focal)
        gen = torch.Generator().manual_seed(int(config.get("init_seed", 0)))
        for key, (shape, dtype) in spec.items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Node())
                node = getattr(node, p)
            t = _init_tensor(key, tuple(shape), dtype, gen)
            if key in _BUFFERS:
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
        self._packed = None
        self._packed_sig = None
        self._weights_epoch = 0
        self._ws: Dict[tuple, dict] = {}
        self._tab: Dict[tuple, torch.Tensor] = {}
        self.fusion_features = None

    # ------------------------------------------------------------------ packing
    # The 16-bit GEMM copies of the weights are rebuilt when the weights change.  Every nn.Module route that
    # replaces or rewrites parameters goes through _apply (.to / .cuda / .float ...) or load_state_dict, which
    # bump an epoch; in-place edits of a parameter (p.mul_(..) under no_grad, nn.init.*_(p), a write through a
    # state_dict() alias) and re-pointed storage (p.data = t) are caught by the per-parameter (version counter,
    # address) pairs that _signature() compares on every call (~0.06 ms of host time for the ~270 tensors of the
    # base model).  The one route torch itself cannot see is an in-place write through ``p.data`` (that view has a
    # version counter of its own): such code calls refresh_weights().
    def _apply(self, fn, *a, **k):
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._weights_epoch += 1
        return super().load_state_dict(*a, **k)

    def refresh_weights(self):
        """Forces the packed 16-bit copies and the captured launch graphs to be rebuilt on the next forward.
        In-place edits of parameters are noticed without it (see _signature); it remains for writes that bypass
        torch's version counter: ``p.data.mul_(..)`` / ``p.data.copy_(..)`` (the ``.data`` view counts versions of
        its own) or a kernel of the caller's writing through ``data_ptr()``."""
        self._weights_epoch += 1

    def _signature(self):
        params = self.__dict__.get("_param_list")
        if params is None or self.__dict__.get("_param_list_epoch") != self._weights_epoch:
            params = [p for p in self.parameters()]
            self.__dict__["_param_list"], self.__dict__["_param_list_epoch"] = params, self._weights_epoch
        return (self.compute_dtype, self._weights_epoch, tuple((p._version, p.data_ptr()) for p in params))

    def _pack(self):
        """16-bit copies of the GEMM weights, fused QK matrices; refreshed whenever a
        parameter was replaced or modified in place (load_state_dict, .to())."""
        if self._packed is not None and self.__dict__.get("_sig_fresh"):
            return self._packed                # checked by the public call that is running (see _forward_split)
        sig = self._signature()
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        dev = sd["backbone.layernorm.weight"].device
        if dev.type != "cuda":
            raise RuntimeError("CognitiveAimModel runs only on a ROCm device (no CPU path): call "
                               ".to('cuda') / create_model(..., device='cuda')")
        _, tdt = _DTYPES[self.compute_dtype]
        D = self.feature_dim
        pk: Dict[str, object] = {"device": dev}
        # The patch embedding takes fp16 operands whatever compute_dtype says: its inputs are normalised pixels
        # and convolution weights, bounded by construction, so the 8-bit significand of bf16 would only cost
        # accuracy (it was the largest single contribution to the bf16 depth error) at the same MFMA rate.
        w = sd["backbone.embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
        kp = (w.shape[1] + 63) // 64 * 64
        wp = torch.zeros(D, kp, device=dev, dtype=torch.float16)
        wp[:, :w.shape[1]] = w.to(torch.float16)
        pk["pe_w"], pk["pe_kp"] = wp, kp
        pk["pe_b"] = sd["backbone.embeddings.patch_embeddings.projection.bias"].float().contiguous()
        pk["cls"] = sd["backbone.embeddings.cls_token"].float().reshape(D).contiguous()
        layers = []
        for i in range(self.num_layers):
            p = f"backbone.encoder.layer.{i}."
            a = p + "attention.attention."
            layers.append({
                "g1": sd[p + "norm1.weight"].float().contiguous(), "b1": sd[p + "norm1.bias"].float().contiguous(),
                "wqkv": torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0)
                .to(tdt).contiguous(),
                "bqkv": torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], 0)
                .float().contiguous(),
                "wo": sd[p + "attention.output.dense.weight"].to(tdt).contiguous(),
                "bo": sd[p + "attention.output.dense.bias"].float().contiguous(),
                "ls1": sd[p + "layer_scale1.lambda1"].float().contiguous(),
                "g2": sd[p + "norm2.weight"].float().contiguous(), "b2": sd[p + "norm2.bias"].float().contiguous(),
                "w1": sd[p + "mlp.fc1.weight"].to(tdt).contiguous(), "bf1": sd[p + "mlp.fc1.bias"].float().contiguous(),
                "w2": sd[p + "mlp.fc2.weight"].to(tdt).contiguous(), "bf2": sd[p + "mlp.fc2.bias"].float().contiguous(),
                "ls2": sd[p + "layer_scale2.lambda1"].float().contiguous(),
            })
        pk["layers"] = layers
        pk["gf"] = sd["backbone.layernorm.weight"].float().contiguous()
        pk["bf"] = sd["backbone.layernorm.bias"].float().contiguous()
        focal = []
        for p in self._focal_prefixes():
            # Row softmax of Q K^T ignores terms that are constant along a row, so
            #   (X Wq^T + bq)(X Wk^T + bk)^T  ~  (X (Wq^T Wk) + bq^T Wk) X^T :
            # ONE 768-wide projection Y = X M^T + v with M = Wk^T Wq, v = bq Wk (softmax scale and log2 e folded
            # in), and the scores are Y X^T against the tokens themselves (src/model.py:192-200).
            wq, wk = sd[p + "query_proj.weight"].float(), sd[p + "key_proj.weight"].float()
            ss = LOG2E / math.sqrt(wq.shape[1] // 8)   # FocalStream.scale = sqrt(patch_dim // num_heads), heads = 8
            # The value projection is followed by the first projection layer with nothing in between (:194, :308-311):
            #   P0 (Wv w + bv s) + b0 = (P0 Wv) w + (P0 bv) s + b0   (s = sum of the column weights),
            # so the [256, D] product is formed once here (in fp64, rounded once) and the per-image chain is two layers.
            p0 = sd[p + "projection.0.weight"].double()
            focal.append({
                "wm": ((wk.t() @ wq) * ss).to(tdt).contiguous(),
                "bv": ((sd[p + "query_proj.bias"].float() @ wk) * ss).contiguous(),
                "m1": (p0 @ sd[p + "value_proj.weight"].double()).float().contiguous(),
                "c1": (p0 @ sd[p + "value_proj.bias"].double()).float().contiguous(),
            })
        pk["focal"] = focal
        pk["f32"] = {k: v.float().contiguous() for k, v in sd.items()
                     if v.dtype.is_floating_point and not k.startswith("backbone.encoder")}
        self._packed, self._packed_sig = pk, sig
        for ws in self._ws.values():          # captured graphs hold the old weight pointers
            ws.pop("graphs", None)
        self._tab.clear()
        # The conversions above were queued on the caller's current stream; forwards run on other streams too
        # (sub-batch streams, forward_async).  Packing is rare, so it simply completes before anything reads it.
        torch.cuda.current_stream(dev).synchronize()
        return pk

    def _focal_prefixes(self):
        """state_dict prefix of every FocalStream in evaluation order."""
        if not self.use_focal:
            return []
        if not self.use_iterative:
            return ["focal_stream."]
        return [f"focal_stream.focal_streams.{i}." for i in range(self.num_iterations)]

    def _table(self, kind: str, *key) -> torch.Tensor:
        k = (kind,) + key
        t = self._tab.get(k)
        if t is None:
            dev = self._packed["device"]
            if kind == "pos":
                t = tables.interpolate_position_embeddings(
                    self.state_dict()["backbone.embeddings.position_embeddings"].detach().cpu(), *key)
            elif kind == "pe":
                t = tables.position_table(*key)
            elif kind == "cbias":
                t = tables.center_bias_table(*key)
            elif kind == "mask":
                t = tables.instruction_mask(*key)
            t = t.to(dev).contiguous()
            self._tab[k] = t
        return t

    def _workspace(self, B: int, gh: int, gw: int) -> dict:
        # one workspace per (shape, stream): calls queued on different streams may overlap
        key = (B, gh, gw, self.compute_dtype, stream_ptr())
        ws = self._ws.get(key)
        if ws is not None:
            self._ws[key] = self._ws.pop(key)          # most recently used last
            return ws
        while len(self._ws) >= self.workspace_cache:
            # evict the least recently used workspace only (with it its captured graphs); the others keep theirs, so
            # a directory of mixed resolutions re-captures the shape that fell out, not every shape
            old = self._ws.pop(next(iter(self._ws)))
            old["stream"].synchronize()
            self._examine_flags(old)
        dev = self._packed["device"]
        _, tdt = _DTYPES[self.compute_dtype]
        D, H = self.feature_dim, self.num_heads
        N = gh * gw
        T = N + 1
        M = B * T
        Tp = (T + 63) // 64 * 64
        ldS = (N + 127) // 128 * 128
        # Grids up to 1536 patches keep the fp32 focal scores of one iteration in memory (every score is then read
        # once for all row statistics); larger grids (120 MB of scores per image at 1036 pixels) never store them:
        # the score tiles are formed three times and reduced in registers (_focal).
        tiled = N > self.scores_in_memory_max_patches
        nblk = call("cogaim_focal_scores_colsum_blocks", N) if tiled else \
            (N + call("cogaim_focal_colsum_rows") - 1) // call("cogaim_focal_colsum_rows")
        parts = call("cogaim_focal_rowstats_parts", N) if tiled else 0
        f32 = dict(device=dev, dtype=torch.float32)
        h16 = dict(device=dev, dtype=tdt)
        big_bytes = max(M * 4 * D * 2, B * N * self._packed["pe_kp"] * 2,
                        B * N * ldS * 4 if (self.use_focal and not tiled) else 0)
        ws = {
            "stream": torch.cuda.current_stream(dev),
            "N": N, "T": T, "M": M, "Tp": Tp, "ldS": ldS, "nblk": nblk, "tiled": tiled, "parts": parts,
            "pstat": torch.empty(2, B * N * parts, **f32) if (tiled and self.use_focal) else None,
            "x": torch.empty(M, D, **f32), "tokens": torch.empty(M, D, **f32),
            "h": torch.empty(M, D, **h16), "qk": torch.empty(M, 3 * D, **h16),     # query | key | value
            "ctx": torch.empty(M, D, **h16),
            "big": torch.empty(big_bytes, device=dev, dtype=torch.uint8),
            "gemm_ws": torch.empty(call("cogaim_gemm_workspace_bytes"), device=dev, dtype=torch.uint8),
            "rmax": torch.empty(B, N, **f32), "rinv": torch.empty(B, N, **f32),
            "part": torch.empty(B * nblk * N, **f32), "attn": torch.empty(B, N, **f32),
            "scale": torch.empty(2, B, N, **f32), "stats": torch.empty(B * ((N + 255) // 256), 6, device=dev, dtype=torch.float64),
            "cand": torch.empty(3, B, N, **f32),
            "flags": torch.zeros(_FLAG_WORDS, device=dev, dtype=torch.int32),
            # host snapshots of the flag words, one per forward, examined when their copy has landed
            "flag_ring": [{"host": torch.zeros(_FLAG_WORDS, dtype=torch.int32).pin_memory(), "event": torch.cuda.Event(),
                           "pending": False} for _ in range(_FLAG_SNAPSHOTS)],
            "flag_next": 0,
            "flags_mapped": False,
            "nchunk": (N + call("cogaim_weighted_tokens_rows") - 1) // call("cogaim_weighted_tokens_rows"),
            "u": torch.empty(B, N, **f32),
            "usum": torch.empty(B, (N + call("cogaim_weighted_tokens_rows") - 1) // call("cogaim_weighted_tokens_rows"), **f32),
            "wt": torch.empty(B, D, **f32), "w": torch.empty(B, D, **f32),
            "wtpart": torch.empty(B * ((N + call("cogaim_weighted_tokens_rows") - 1)
                                       // call("cogaim_weighted_tokens_rows")) * D, **f32),
            "t256": torch.empty(B, 256 * max(1, self.num_iterations), **f32), "a256": torch.empty(B, 256, **f32), "t128": torch.empty(B, 128, **f32), "t64": torch.empty(B, 64, **f32),
            "feats": torch.empty(B, 64 * max(1, self.num_iterations), **f32),
            "exf": torch.empty(B, 128, **f32), "cont": torch.empty(B, 3, **f32),
            "cat": torch.zeros(B, 192, **f32), "fused": torch.empty(B, 192, **f32),
            "guided": torch.empty(B, N, **f32), "out": torch.empty(2, B, **f32),
            "pw": torch.empty(64, D, **f32), "pb": torch.empty(64, **f32),
        }
        # may a kernel store the flag words straight into the pinned snapshots?  (page-locked memory is mapped into the
        # device's address space at its host address on this platform; checked, not assumed)
        ws["flags_mapped"] = all(call("cogaim_host_pointer_ok", sl["host"].data_ptr()) == 0 for sl in ws["flag_ring"])
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ kernels
    def _linear(self, x, ldx, wkey, out, ldo, B, act=0, W=None, bias=None, bias2=None, bias2_scale=None, scale_parts=1):
        f = self._packed["f32"]
        W = f[wkey + ".weight"] if W is None else W
        bias = f[wkey + ".bias"] if bias is None else bias
        O, K = W.shape
        call("cogaim_linear_f32", x, ldx, ptr(W), ptr(bias), ptr(bias2), ptr(bias2_scale), scale_parts, out, ldo, B, O, K,
             act, stream_ptr())

    def _backbone(self, images: Optional[torch.Tensor], shape: Tuple[int, int, int], ws: dict):
        """DINOv2 forward (hf:451-477) -> ws['tokens'] fp32 [B*T, D] (final LayerNorm applied).
        Two parts: with ``images`` only the patch gather into the workspace runs; with ``images=None`` the
        rest (patch embedding GEMM ... final LayerNorm), which touches workspace buffers only."""
        pk, st = self._packed, stream_ptr()
        dt, _ = _DTYPES[self.compute_dtype]
        B, Hi, Wi = shape
        p, D, H = self.patch_size, self.feature_dim, self.num_heads
        gh, gw = Hi // p, Wi // p
        N, T, M, Tp = ws["N"], ws["T"], ws["M"], ws["Tp"]
        pos = self._table("pos", gh, gw)
        x, h, qk, ctx, big = (ws[k].data_ptr() for k in ("x", "h", "qk", "ctx", "big"))
        gw, gwb = ws["gemm_ws"].data_ptr(), ws["gemm_ws"].numel()
        rf = ws["flags"].data_ptr() + 4 * _FLAG_RANGE
        if images is not None:                 # eager: reads the caller's tensor (everything after it is static)
            call("cogaim_im2col", images.data_ptr(), big, B, Hi, Wi, p, pk["pe_kp"], 1, st)
            return
        call("cogaim_cls_row", ptr(pk["cls"]), ptr(pos), x, B, T, D, st)
        call("cogaim_gemm_patch_embed", big, ptr(pk["pe_w"]), ptr(pk["pe_b"]), ptr(pos), x, B, N, D, pk["pe_kp"], 1, gw, gwb, st)
        qscale = 64 ** -0.5 * LOG2E
        eps = self.layer_norm_eps
        for L in pk["layers"]:
            call("cogaim_layernorm", x, ptr(L["g1"]), ptr(L["b1"]), h, M, D, eps, dt, st)
            # one projection GEMM for query | key | value (the query columns carry the softmax scale)
            call("cogaim_gemm_bias_half", h, ptr(L["wqkv"]), ptr(L["bqkv"]), qk, M, 3 * D, D, D, D, 3 * D, D, qscale, dt, gw, gwb, rf, st)
            call("cogaim_mhsa_qkv_fwd", qk, ctx, B, T, H, 3 * D, dt, st)
            if self.fuse_out_proj_layernorm and D == 768:
                call("cogaim_gemm_residual_layernorm", ctx, ptr(L["wo"]), ptr(L["bo"]), ptr(L["ls1"]), x, ptr(L["g2"]),
                     ptr(L["b2"]), h, M, D, D, D, D, D, D, eps, dt, st)
            else:
                call("cogaim_gemm_residual_f32", ctx, ptr(L["wo"]), ptr(L["bo"]), ptr(L["ls1"]), x, M, D, D, D, D, D, dt, gw, gwb, st)
                call("cogaim_layernorm", x, ptr(L["g2"]), ptr(L["b2"]), h, M, D, eps, dt, st)
            call("cogaim_gemm_bias_gelu_half", h, ptr(L["w1"]), ptr(L["bf1"]), big, M, 4 * D, D, D, D, 4 * D, dt, gw, gwb, rf, st)
            call("cogaim_gemm_residual_f32", big, ptr(L["w2"]), ptr(L["bf2"]), ptr(L["ls2"]), x, M, D, 4 * D, 4 * D, 4 * D, D, dt, gw, gwb, st)
        call("cogaim_layernorm", x, ptr(pk["gf"]), ptr(pk["bf"]), ws["tokens"].data_ptr(), M, D, eps, 2, st)

    def _focal(self, B: int, ws: dict):
        """IterativeFocalStream (src/model.py:391-455) on ws['tokens'] -> ws['cat'][:, 64:128], ws['attn']."""
        pk, st = self._packed, stream_ptr()
        dt, _ = _DTYPES[self.compute_dtype]
        D, N, T, ldS = self.feature_dim, ws["N"], ws["T"], ws["ldS"]
        tok = ws["tokens"].data_ptr() + D * 4          # skip the CLS row of every image
        tstride = T * D
        pe, cb = self._table("pe", N, D), self._table("cbias", N)
        h, qk, S = ws["h"].data_ptr(), ws["qk"].data_ptr(), ws["big"].data_ptr()
        I = self.num_iterations
        for i in range(I):
            f = pk["focal"][i]
            pfx = self._focal_prefixes()[i]
            s_in = None if i == 0 else ws["scale"][i % 2].data_ptr()
            s_out = None if i == I - 1 else ws["scale"][(i + 1) % 2].data_ptr()
            call("cogaim_focal_prep", tok, tstride, s_in, ptr(pe), h, B, N, D, dt, st)
            # Y = X~ M^T + v into the first D columns' worth of the qk buffer, then S[b] = Y[b] X~[b]^T (log2 domain)
            call("cogaim_gemm_bias_half", h, ptr(f["wm"]), ptr(f["bv"]), qk, B * N, D, D, D, D, D, 0, 1.0, dt,
                 ws["gemm_ws"].data_ptr(), ws["gemm_ws"].numel(), ws["flags"].data_ptr() + 4 * _FLAG_RANGE, st)
            if ws["tiled"]:
                # no score matrix in memory: S tiles are formed three times and reduced in registers
                pm, pl = ws["pstat"][0].data_ptr(), ws["pstat"][1].data_ptr()
                call("cogaim_focal_scores_rowstats", qk, h, pm, pl, N, D, D, N * D, B, dt, st)
                call("cogaim_focal_rowstats_combine", pm, pl, ws["parts"], ptr(ws["rmax"]), ptr(ws["rinv"]), B, N, st)
                call("cogaim_focal_scores_colsum", qk, h, ptr(ws["rmax"]), ptr(ws["rinv"]), None, ptr(ws["part"]), N, D, D,
                     N * D, B, dt, st)
            else:
                call("cogaim_gemm_batched_f32", qk, h, S, N, N, D, D, D, ldS, N * D, N * D, N * ldS, B, dt, st)
                call("cogaim_focal_softmax_stats_colsum", S, ldS, N * ldS, ptr(ws["rmax"]), ptr(ws["rinv"]), ptr(ws["part"]), B, N, st)
            # :234-282: column mean / diagonal / row maximum (+ centre bias) and the reference's variance cascade
            call("cogaim_focal_attention_candidates", ptr(ws["part"]), ws["nblk"], ptr(cb), None if ws["tiled"] else S, ldS,
                 N * ldS, ptr(ws["rmax"]), ptr(ws["rinv"]), ptr(ws["cand"]), ptr(ws["stats"]), B, N, st)
            call("cogaim_focal_attention_vector", ptr(ws["cand"]), ptr(ws["stats"]), 0 if ws["tiled"] else 1, ptr(ws["attn"]),
                 s_in, s_out, self.focus_strength, B, N, ptr(ws["flags"]), i, st)
            if ws["tiled"]:
                call("cogaim_focal_scores_colsum", qk, h, ptr(ws["rmax"]), ptr(ws["rinv"]), ptr(ws["attn"]), ptr(ws["part"]),
                     N, D, D, N * D, B, dt, st)
            else:
                call("cogaim_focal_weighted_colsum", S, ldS, N * ldS, ptr(ws["rmax"]), ptr(ws["rinv"]), ptr(ws["attn"]), ptr(ws["part"]), B, N, st)
            # u = a^T P is reduced from its partials inside the token pass; its sum per chunk of rows goes to ws['usum']
            call("cogaim_weighted_tokens_parts", tok, tstride, s_in, ptr(pe), ptr(ws["part"]), ws["nblk"], ptr(ws["wt"]),
                 ptr(ws["wtpart"]), ptr(ws["usum"]), B, N, D, st)
            # value projection folded into projection.0 (see _pack) + ReLU: wide enough for one wave per output;
            # projection.3 of every iteration runs inside the fusion chain (_focal_fused_into)
            self._linear(ptr(ws["wt"]), D, pfx + "projection.0", ws["t256"].data_ptr() + 256 * i * 4, 256 * I, B, act=1,
                         W=f["m1"], bias2=f["c1"], bias2_scale=ws["usum"], scale_parts=ws["nchunk"])
        return ws["attn"]

    def _focal_fused_into(self, ws, B, col):
        w32, I = self._packed["f32"], self.num_iterations
        pfx = self._focal_prefixes()
        cat = ws["cat"].data_ptr() + col * 4
        # projection.3 of each iteration (256 -> 64, :82-87), then the fusion MLP over their concatenation (:342-346, 430)
        # -- a bare FocalStream returns its projected features as they are (:311).  A chain holds at most eight layers:
        # with more than six iterations each goes through a launch of its own that leaves its features in ws['feats'].
        p3 = lambda i, src, dst: (ptr(w32[pfx[i] + "projection.3.weight"]), ptr(w32[pfx[i] + "projection.3.bias"]), None,
                                  256, 64, 0, src, dst)
        if not self.use_iterative:
            mlp_chain([p3(0, 0, 1024)], ptr(ws["t256"]), 256, 256, cat, 192, 64, 1024, B)
            return
        fus = lambda src, mid, dst: [
            (ptr(w32["focal_stream.fusion.0.weight"]), ptr(w32["focal_stream.fusion.0.bias"]), None, 64 * I, 128, 1, src, mid),
            (ptr(w32["focal_stream.fusion.2.weight"]), ptr(w32["focal_stream.fusion.2.bias"]), None, 128, 64, 0, mid, dst)]
        if I <= 6:       # t256 rows [B, 256 I] at 0; features at 2048; fusion hidden at 3072; result at 3584
            mlp_chain([p3(i, 256 * i, 2048 + 64 * i) for i in range(I)] + fus(2048, 3072, 3584),
                      ptr(ws["t256"]), 256 * I, 256 * I, cat, 192, 64, 3584, B)
            return
        for i in range(I):       # one launch per iteration (256 -> 64) ...
            mlp_chain([p3(i, 0, 1024)], ws["t256"].data_ptr() + 256 * i * 4, 256 * I, 256,
                      ws["feats"].data_ptr() + 64 * i * 4, 64 * I, 64, 1024, B)
        mlp_chain(fus(0, 64 * I, 64 * I + 128), ptr(ws["feats"]), 64 * I, 64 * I, cat, 192, 64, 64 * I + 128, B)       # ... and the fusion

    def _check_fallback_flags(self, ws):
        """The reference switches to other attention summaries when the whole-batch variance of (column mean +
        centre bias) is below 1e-6: the diagonal, then the row maximum, then token norms plus noise
        (src/model.py:242-257).  The first two steps of that cascade are evaluated on the device inside the forward
        (they can only trigger on grids of 2, 3 or 2 x 2 patches: the constant the sum would have to equal must reach
        the bias at the centre, 0.3); the noise step is not reproduced and is reported through the flag words.

        Every forward copies the flag words to one of a few pinned host snapshots asynchronously; a snapshot is
        examined by a later call on this workspace once its copy has landed, or by ``synchronize()``: the forward
        never waits for the device (waiting for the previous forward's snapshot, as a single snapshot would need,
        leaves the GPU idle between two forwards until the host has issued the next one).  A caller that wants these
        conditions raised at a definite point calls ``model.synchronize()``.  An out-of-range camera index
        additionally makes that image's depth and confidence NaN, so its result cannot pass for a prediction."""
        slot = self._flag_slot(ws)
        slot["host"].copy_(ws["flags"], non_blocking=True)
        self._flag_slot_filled(ws, slot)

    def _flag_slot(self, ws):
        """The snapshot the running call fills (examines the older ones first)."""
        self._examine_flags(ws, wait=False)
        slot = ws["flag_ring"][ws["flag_next"]]
        if slot["pending"]:                    # the host is a whole ring of forwards ahead of the device
            self._examine_flags(ws, wait=True)
        return slot

    def _flag_slot_filled(self, ws, slot):
        slot["event"].record()
        slot["pending"] = True
        ws["flag_next"] = (ws["flag_next"] + 1) % _FLAG_SNAPSHOTS

    def _new_outputs(self, B, N, device):
        """Fresh result tensors of one call, carved out of ONE allocation: depth [B,1], conf [B,1], fused [B,192],
        attention [B,N] (None without a focal stream); each is contiguous."""
        n_attn = N if self.use_focal else 0
        flat = torch.empty(B * (2 + self.fusion_dim + n_attn), device=device, dtype=torch.float32)
        depth, conf = flat[:B].view(B, 1), flat[B:2 * B].view(B, 1)
        fused = flat[2 * B: 2 * B + B * self.fusion_dim].view(B, self.fusion_dim)
        attn = flat[2 * B + B * self.fusion_dim:].view(B, N) if n_attn else None
        return depth, conf, attn, fused

    def _export(self, ws, B, dst, attn_src):
        """Copies depth / confidence / fused features / attention out of the workspace into `dst` (the caller's
        tensors) and takes the flag snapshot, in one launch (no per-tensor clones or copies)."""
        depth, conf, attn, fused = dst
        slot = self._flag_slot(ws)
        mapped = ws["flags_mapped"]
        out = ws["out"]
        call("cogaim_export_outputs", out.data_ptr(), out.data_ptr() + B * 4, ptr(ws["fused"]), self.fusion_dim,
             ptr(attn_src) if attn is not None else None, ws["N"], ptr(depth), ptr(conf), ptr(fused), ptr(attn),
             ptr(ws["flags"]), slot["host"].data_ptr() if mapped else None, _FLAG_WORDS, B, stream_ptr())
        if not mapped:
            slot["host"].copy_(ws["flags"], non_blocking=True)
        self._flag_slot_filled(ws, slot)

    def _examine_flags(self, ws, wait=True):
        """Oldest snapshot first; without `wait`, stops at the first one whose copy has not landed yet."""
        ring, first = ws["flag_ring"], ws["flag_next"]
        for j in range(_FLAG_SNAPSHOTS):
            slot = ring[(first + j) % _FLAG_SNAPSHOTS]
            if not slot["pending"]:
                continue
            if wait:
                slot["event"].synchronize()
            elif not slot["event"].query():
                return
            slot["pending"] = False
            host = slot["host"]
            err = None
            if int(host[_FLAG_CAMERA]):
                ws["flags"][_FLAG_CAMERA] = 0
                err = IndexError("camera_idx out of range in an earlier forward (index out of range in self); "
                                 "depth and confidence of that image are NaN")
            elif int(host[_FLAG_RANGE]):
                ws["flags"][_FLAG_RANGE] = 0
                err = OverflowError("an activation exceeded the fp16 range (65504) in an earlier forward and was "
                                    "stored as inf: results of that call are not valid.  Build the model with "
                                    "compute_dtype='bf16' (same speed, fp32 exponent range) for these weights")
            elif ws["N"] > 1 and bool(host[: self.num_iterations].any()):
                err = RuntimeError("in an earlier forward the focal attention summaries (column mean, diagonal, row "
                                   "maximum: src/model.py:234-249) were all degenerate (whole-batch variance < 1e-6); "
                                   "the reference then draws noise (:252-257), which is not reproduced")
            if err is not None:
                # reported once: younger snapshots were taken before the device words were cleared just above
                for other in ring:
                    if other["pending"]:
                        other["event"].synchronize()
                        other["pending"] = False
                raise err

    def synchronize(self):
        """Waits for all queued work of this model and raises any deferred device-side condition."""
        torch.cuda.synchronize()
        for ws in self._ws.values():
            self._examine_flags(ws)

    def _ambient(self, ws, B, col):
        T, D = ws["T"], self.feature_dim
        cls = ws["tokens"].data_ptr()                  # row b*T is the CLS token of image b
        w, a = self._packed["f32"], "ambient_stream.mlp."           # src/model.py:37-44, one launch
        # the 768-wide first layer with one wave per output (fills the chip); the two narrow layers in one launch
        self._linear(cls, T * D, a + "0", ptr(ws["a256"]), 256, B, act=1)
        mlp_chain([(ptr(w[a + "3.weight"]), ptr(w[a + "3.bias"]), None, 256, 128, 1, 0, 1024),
                   (ptr(w[a + "5.weight"]), ptr(w[a + "5.bias"]), None, 128, 64, 0, 1024, 2048)],
                  ptr(ws["a256"]), 256, 256, ws["cat"].data_ptr() + col * 4, 192, 64, 2048, B)

    @staticmethod
    def _exif_check(exif_data, B: int):
        """Host-only validation of an EXIF dict (nothing is launched, nothing copied): the four vectors of
        src/model.py:491-508 are present, are tensors and hold one entry per image."""
        if not isinstance(exif_data, dict):
            raise TypeError(f"exif_data must be a dict of tensors, got {type(exif_data).__name__}")
        for k in ("camera_idx", "focal_length", "aperture", "iso"):
            t = exif_data[k]                                  # KeyError, as in the reference
            if not torch.is_tensor(t):
                raise TypeError(f"exif_data['{k}'] must be a tensor")
            if t.numel() != B:
                raise ValueError(f"exif_data['{k}'] has {t.numel()} entries for a batch of {B}")

    def _exif_inputs(self, exif_data: dict, B: int):
        """Validated device copies of the four EXIF vectors ([B] or [B,1], src/model.py:491-508)."""
        dev = self._packed["device"]

        def vec(k, dtype):
            t = exif_data[k]
            if t.dim() > 1:
                t = t.squeeze(1)
            t = t.to(device=dev, dtype=dtype).contiguous()
            if t.numel() != B:
                raise ValueError(f"exif_data['{k}'] has {t.numel()} entries for a batch of {B}")
            return t
        idx = vec("camera_idx", torch.int64)
        return idx, vec("focal_length", torch.float32), vec("aperture", torch.float32), vec("iso", torch.float32)

    def _exif_prep(self, exif_in, ws, B):
        """Embedding gather + [focal, aperture, log(iso+1)] (src/model.py:491-512): the only EXIF step that
        reads the caller's tensors."""
        idx, fl, ap, iso = exif_in
        emb = self._packed["f32"]["exif_prior.camera_embedding.weight"]
        # an out-of-range camera index makes the image's outputs NaN and is reported with the other flags
        call("cogaim_exif_prep", ptr(idx), ptr(fl), ptr(ap), ptr(iso), ptr(emb), emb.shape[0], 64, ptr(ws["exf"]), 128,
             ptr(ws["cont"]), B, ws["flags"].data_ptr() + _FLAG_CAMERA * 4, stream_ptr())

    def _exif(self, ws, B, col):
        # src/model.py:510-519 in one launch: encoder(3 -> 64 -> 64) on the continuous values, its output behind the
        # camera embedding (columns 0..63 of ws['exf'], gathered by _exif_prep), fusion(128 -> 256 -> 64)
        w, e = self._packed["f32"], "exif_prior."
        mlp_chain([(ptr(w[e + "exif_encoder.0.weight"]), ptr(w[e + "exif_encoder.0.bias"]), None, 3, 64, 1, 0, 2048),
                   (ptr(w[e + "exif_encoder.2.weight"]), ptr(w[e + "exif_encoder.2.bias"]), None, 64, 64, 0, 2048, 1024 + 64),
                   (ptr(w[e + "fusion.0.weight"]), ptr(w[e + "fusion.0.bias"]), None, 128, 256, 1, 1024, 3072),
                   (ptr(w[e + "fusion.3.weight"]), ptr(w[e + "fusion.3.bias"]), None, 256, 64, 0, 3072, 0)],
                  ptr(ws["cont"]), 3, 3, ws["cat"].data_ptr() + col * 4, 192, 64, 0, B,
                  in1=ptr(ws["exf"]), in1_ld=128, in1_len=64, in1_off=1024)

    def _heads(self, ws, B, fuse=True):
        f = self._packed["f32"]
        out = ws["out"]
        if fuse:        # fusion Linear(192, 192) + ReLU and both heads, one launch
            call("cogaim_fusion_heads", ptr(ws["cat"]), 192, 192, ptr(f["fusion.0.weight"]), ptr(f["fusion.0.bias"]), 192,
                 ptr(f["decision_head.0.weight"]), ptr(f["decision_head.0.bias"]),
                 ptr(f["confidence_head.0.weight"]), ptr(f["confidence_head.0.bias"]),
                 ptr(f["confidence_head.2.weight"]), ptr(f["confidence_head.2.bias"]),
                 ptr(ws["fused"]), out.data_ptr(), out.data_ptr() + B * 4, B, stream_ptr())
            return
        call("cogaim_heads", ptr(ws["fused"]), 192, ptr(f["decision_head.0.weight"]), ptr(f["decision_head.0.bias"]),
             ptr(f["confidence_head.0.weight"]), ptr(f["confidence_head.0.bias"]),
             ptr(f["confidence_head.2.weight"]), ptr(f["confidence_head.2.bias"]),
             out.data_ptr(), out.data_ptr() + B * 4, B, stream_ptr())

    def _empty_outputs(self, images, return_attention):
        z = torch.zeros(0, 1, device=images.device, dtype=torch.float32)
        n = (images.shape[2] // self.patch_size) * (images.shape[3] // self.patch_size)
        attn = torch.zeros(0, n, device=images.device) if self.use_focal else None
        return (z, z.clone(), attn) if return_attention else (z, z.clone())

    def _outputs(self, ws):
        out = ws["out"].clone()
        return out[0].unsqueeze(1), out[1].unsqueeze(1)

    def _prepare(self, images: torch.Tensor):
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in "
                             f"the configuration. Expected 3 but got {images.shape[1] if images.dim() == 4 else '?'}.")
        self._pack()
        dev = self._packed["device"]
        if images.device != dev:
            raise RuntimeError(f"images are on {images.device}, model on {dev}")
        images = images.to(torch.float32).contiguous()
        B, _, Hi, Wi = images.shape
        gh, gw = Hi // self.patch_size, Wi // self.patch_size
        if gh < 1 or gw < 1:
            raise ValueError("image smaller than one patch")
        return images, self._workspace(B, gh, gw), B

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def get_features_aligned(self, images, exif_data=None):
        """Fused 192-d features (reference src/model.py:960-1056), one pass."""
        if exif_data is not None and self.use_exif and images.dim() == 4:
            try:
                self._exif_check(exif_data, images.shape[0])
            except Exception as e:
                if self.strict_errors:
                    raise
                print(f"get_features_aligned exception: {e!r}")
                return torch.zeros(images.shape[0], self.fusion_dim, device=images.device)     # :1054
        images, ws, B = self._prepare(images)
        self._features(images, exif_data, ws, B, graph_key="features")
        self._check_fallback_flags(ws)
        return ws["fused"].clone()

    def _features(self, images, exif_data, ws, B, focal_override=None, heads=True, graph_key=None, tail=None):
        """Fills ws['cat'] with [ambient | focal | exif] (zero where a module is absent), then (``heads``)
        ws['fused'] and ws['out'].  Returns the number of feature columns filled.

        The two launches that read the caller's tensors (patch gather, EXIF gather) run eagerly; the rest of
        the sequence touches only workspace buffers and weights, so after its first eager run it is captured
        once per (workspace, variant) and replayed as one HIP graph."""
        if not (self.use_ambient or self.use_focal or (self.use_exif and exif_data is not None)):
            raise RuntimeError("Cognitive modules returned no features!")
        exif_in = self._exif_inputs(exif_data, B) if (self.use_exif and exif_data is not None) else None
        shape = (B, images.shape[2], images.shape[3])
        self._backbone(images, shape, ws)
        if exif_in is not None:
            self._exif_prep(exif_in, ws, B)
        ncols = 64 * (int(self.use_ambient) + int(self.use_focal) + int(exif_in is not None))

        def body():
            self._backbone(None, shape, ws)
            if ncols != 192:               # columns of absent modules are zero (:1032-1039); present ones are all rewritten
                ws["cat"].zero_()
            col_a, col_f = 0, 64 * int(self.use_ambient)
            col_e = col_f + 64 * int(self.use_focal)
            # (The ambient and EXIF MLPs depend on the backbone only, but running them on a forked stream beside the focal
            # stream -- a branch of the captured graph -- made the step 1.1 ms SLOWER: 13.88 against 12.73 ms on one box.)
            if self.use_ambient:
                self._ambient(ws, B, col_a)
            if self.use_focal:
                self._focal(B, ws)
                if focal_override is None:
                    self._focal_fused_into(ws, B, col_f)
                else:
                    focal_override(col_f)
            if exif_in is not None:
                self._exif(ws, B, col_e)
            if heads:           # absent modules leave zero columns, as the reference's zero padding does (:1032-1045)
                self._heads(ws, B)
            if tail is not None:
                tail()

        self._run(ws, (graph_key, exif_in is not None, heads), body)
        return ncols

    def _run(self, ws, key, body):
        """Eager the first time a (workspace, variant) is seen -- which also loads every kernel -- then
        captured into a HIP graph and replayed.  Per-launch profiling and ``use_graph=False`` stay eager."""
        from . import _lib
        if not self.use_graph or key[0] is None or _lib.PROFILE is not None:
            body()
            return
        graphs = ws.setdefault("graphs", {})
        entry = graphs.get(key)
        if entry is None:
            body()
            graphs[key] = "seen"
        elif entry == "seen":
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            # thread_local: other host threads (the batch loop's uploader) keep issuing copies, allocations and event
            # waits on their own streams while this thread records; only this thread's calls belong to the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
            graphs[key] = g
            g.replay()
        else:
            entry.replay()

    def get_features(self, images, exif_data=None):
        return self.get_features_aligned(images, exif_data)

    def get_attention_weights(self):
        return getattr(self, "_last_attention_weights", None)

    # ------------------------------------------------------------------ curiosity side effects (opt-in)
    def _curiosity_draw(self, B):
        """The two draws of one CuriosityModule.forward, in its order, from torch's CPU generator (the reference
        run on a CPU draws there; src/model.py:609 then :744)."""
        D = self.feature_dim
        eps = torch.randn(B, D // 4)
        noise = torch.randn(B, D) * 0.01
        return eps, noise

    def _curiosity_run(self, ws, B, draw):
        """CuriosityModule.forward on the CLS tokens of ws['tokens'] (src/model.py:586-688, exif_data=None as at every
        call site of the forward paths) + the ring-buffer update (:760-773), all on the device."""
        D, T = self.feature_dim, ws["T"]
        L, st = D // 4, stream_ptr()
        dev = self._packed["device"]
        cw = ws.get("curiosity")
        if cw is None:
            f32 = dict(device=dev, dtype=torch.float32)
            cw = ws["curiosity"] = {k: torch.empty(B, n, **f32) for k, n in (
                ("eps", L), ("noise", D), ("noisy", D), ("h384", D // 2), ("mu", L), ("logvar", L), ("z", L), ("rec", L),
                ("h192", L), ("h128", 128), ("unc", 1), ("lb", 1), ("ln", 1), ("reward", 1))}
        cw["eps"].copy_(draw[0]); cw["noise"].copy_(draw[1])
        cls, ld = ws["tokens"].data_ptr(), T * D
        c, lin, p = "curiosity_module.", self._linear, (lambda k: cw[k].data_ptr())
        lin(cls, ld, c + "encoder_mean.0", p("h384"), D // 2, B, act=1)
        lin(p("h384"), D // 2, c + "encoder_mean.3", p("mu"), L, B)
        lin(cls, ld, c + "encoder_logvar.0", p("h384"), D // 2, B, act=1)
        lin(p("h384"), D // 2, c + "encoder_logvar.3", p("logvar"), L, B)
        call("cogaim_curiosity_sample", p("mu"), p("logvar"), p("eps"), p("z"), cls, ld, p("noise"), p("noisy"), B, L, D, st)
        lin(p("z"), L, c + "decoder.0", p("h384"), D // 2, B, act=1)
        lin(p("h384"), D // 2, c + "decoder.3", p("rec"), L, B)
        lin(cls, ld, c + "uncertainty_head.0", p("h192"), L, B, act=1)
        lin(p("h192"), L, c + "uncertainty_head.2", p("unc"), 1, B)
        lin(cls, ld, c + "local_curiosity.0", p("h128"), 128, B, act=1)
        lin(p("h128"), 128, c + "local_curiosity.2", p("lb"), 1, B)
        lin(p("noisy"), D, c + "local_curiosity.0", p("h128"), 128, B, act=1)
        lin(p("h128"), 128, c + "local_curiosity.2", p("ln"), 1, B)
        hist, hp = self.curiosity_module.exploration_history, self.curiosity_module.history_pointer
        if hist.device != dev or hp.device != dev or hist.dtype != torch.float32 or hp.dtype != torch.int64:
            raise RuntimeError("curiosity buffers must live on the model's device (fp32 history, int64 pointer)")
        call("cogaim_curiosity_finish", p("rec"), cls, ld, p("mu"), p("logvar"), p("unc"), p("lb"), p("ln"),
             ptr(self._packed["f32"][c + "curiosity_weights"]), p("reward"), hist.data_ptr(), hp.data_ptr(), B, L,
             hist.numel(), st)
        return cw["reward"]

    def _forward_core(self, images, exif_data, curiosity_calls=0, zero_features=False, dst=None):
        images, ws, B = self._prepare(images)
        draws = [self._curiosity_draw(B) for _ in range(curiosity_calls)]
        if zero_features:
            # the reference's backup features (src/model.py:1050-1056): depth and confidence come from all-zero
            # fused features, the attention map (:1093-1113) still from the images
            if self.use_ambient or self.use_focal:
                self._features(images, None, ws, B, graph_key="forward")
            ws["fused"].zero_()
            self._heads(ws, B, fuse=False)
        else:
            self._features(images, exif_data, ws, B, graph_key="forward")
        for dr in draws:
            self._curiosity_run(ws, B, dr)
        if dst is None:
            dst = self._new_outputs(B, ws["N"], images.device)
        self._export(ws, B, dst, ws["attn"] if self.use_focal else None)
        return dst

    def _forward_split(self, images, exif_data, n, zero_features=False):
        """The batch as n contiguous sub-batches, each on a stream of the model's own (its own workspace and
        captured graph); the caller's stream waits for all of them."""
        B = images.shape[0]
        dev = images.device
        if self._side_streams is None or len(self._side_streams) < n or self._side_streams[0].device != dev:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        cur = torch.cuda.current_stream(dev)
        bounds = [(B * i // n, B * (i + 1) // n) for i in range(n)]
        p = self.patch_size
        # the whole batch's result tensors, allocated on the caller's stream; every sub-batch writes its rows
        whole = self._new_outputs(B, (images.shape[2] // p) * (images.shape[3] // p), dev)
        self._pack()                           # on the caller's stream, once for all sub-batches
        self.__dict__["_sig_fresh"] = True
        try:
            for st, (lo, hi) in zip(self._side_streams, bounds):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    ex = None if exif_data is None else {k: v[lo:hi] for k, v in exif_data.items()}
                    self._forward_core(images[lo:hi], ex, zero_features=zero_features,
                                       dst=tuple(None if t is None else t[lo:hi] for t in whole))
        finally:
            self.__dict__["_sig_fresh"] = False
        for st in self._side_streams[:n]:
            cur.wait_stream(st)
        return whole

    @torch.no_grad()
    def forward(self, images, exif_data=None, return_attention=False):
        """(depth [B,1], confidence [B,1]) or (+ attention [B,N] | None)  — reference src/model.py:1064-1155."""
        if images.dim() == 4 and images.shape[0] == 0:          # empty batch: empty results, nothing launched
            return self._empty_outputs(images, return_attention)
        n = min(self.num_streams, images.shape[0] // 8) if images.dim() == 4 and images.is_cuda else 1
        calls = 0
        if self.faithful_curiosity and self.use_focal:        # :995, :1104 (first call only), :1139
            calls = 1 + int(not hasattr(self, "_last_attention_weights")) + int(bool(return_attention))
            n = 1                                             # whole-batch draws, in the reference's order
        zero = False
        if exif_data is not None and self.use_exif and images.dim() == 4:
            try:
                self._exif_check(exif_data, images.shape[0])
            except Exception as e:
                if self.strict_errors:
                    raise
                print(f"get_features_aligned exception: {e!r}")      # the reference's message (:1051)
                zero, exif_data = True, None
        if n > 1:
            depth, conf, attn, fused = self._forward_split(images, exif_data, n, zero)
        else:
            depth, conf, attn, fused = self._forward_core(images, exif_data, calls, zero)
        self.fusion_features = fused
        if self.use_focal and not hasattr(self, "_last_attention_weights"):
            self._last_attention_weights = attn          # :1093-1113
        elif not self.use_focal:
            self._last_attention_weights = None          # :1116-1117
        if return_attention:
            return depth, conf, attn
        return depth, conf

    @torch.no_grad()
    def forward_async(self, images, exif_data=None, return_attention=False):
        """``forward`` for loops that can keep two batches in flight (no counterpart in the reference).

        Returns at once with a handle; ``handle.result()`` gives what ``forward`` returns and makes the caller's
        current stream wait for it.  Calls alternate between two streams of the model's own, each running the WHOLE
        batch (its GEMMs at full height), so the focal / tail part of one batch lies under the backbone of the next:
        ``forward`` cannot do that, because its outputs have to be valid on the caller's stream when it returns and
        the next call's inputs are only known to be ready at that stream's tail.  The caller keeps ``images`` and
        ``exif_data`` alive and unchanged until ``result()``; results are collected in call order."""
        if self.faithful_curiosity:
            raise RuntimeError("forward_async keeps batches in flight out of order with the host's random stream; "
                               "faithful_curiosity needs forward()")
        if not (images.dim() == 4 and images.is_cuda) or images.shape[0] == 0:
            return _Pending(self, None, None, None, return_attention, ready=self.forward(images, exif_data, return_attention))
        dev = images.device
        slots = getattr(self, "_async_streams", None)
        if slots is None or slots[0].device != dev:
            slots = self._async_streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            self._async_turn = 0
        st = slots[self._async_turn]
        self._async_turn ^= 1
        self._pack()                           # on the caller's stream: the model's own streams only read the copies
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)                      # the inputs as the caller's stream has them NOW
        st.wait_event(ready)
        images.record_stream(st)
        with torch.cuda.stream(st):
            outs = self._forward_core(images, exif_data)
            done = torch.cuda.Event()
            done.record(st)
        return _Pending(self, outs, done, st, return_attention)

    @torch.no_grad()
    def forward_with_guidance(self, images, exif_data=None, attention_guidance=None, return_attention=False,
                              temp_projection: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """Instruction-guided forward (reference src/model.py:1157-1240).

        ``temp_projection=(weight [64,768], bias [64])`` pins the Linear(768,64) that the
        reference re-creates with fresh random weights on every guided call (:1417-1422);
        when omitted a new ``nn.Linear(768, 64)`` is drawn from torch's global RNG exactly
        as the reference does, so guided depth is as irreproducible here as it is there.

        ``attention_guidance`` may also be a list of K instructions (strings or tensors): the
        backbone and the focal stream, which do not depend on the instruction (:1257), run once
        and the K masks are applied together.  The result is what K single calls in list order
        return, stacked: depth [K,B,1], confidence [K,B,1] (and attention [K,B,N]);
        ``temp_projection`` is then one pair for all, or a list of K pairs.
        """
        if isinstance(attention_guidance, (list, tuple)):
            return self._guided_sweep(images, exif_data, list(attention_guidance), return_attention, temp_projection)
        if images.dim() == 4 and images.shape[0] == 0:
            return self._empty_outputs(images, return_attention)
        if not self.use_focal or attention_guidance is None:
            if self.use_focal or self.use_ambient:
                out = self.forward(images, exif_data, True)
                if self.use_focal:
                    self._last_attention_weights = out[2]
                return out if return_attention else out[:2]
            return self.forward(images, exif_data, return_attention)
        exif_bad = None
        try:            # everything in here is host work: nothing has been launched when it raises
            p = self.patch_size
            if images.dim() == 4:
                g = tables.guidance_vector(attention_guidance, (images.shape[2] // p) * (images.shape[3] // p))
                if g.numel() != (images.shape[2] // p) * (images.shape[3] // p):
                    raise ValueError("attention_guidance cannot be resized to the patch grid")
            if exif_data is not None and self.use_exif and images.dim() == 4:
                try:
                    self._exif_check(exif_data, images.shape[0])
                except Exception as e:
                    if self.strict_errors:
                        raise
                    # the reference fails on the EXIF dict only after the guided attention has been formed and
                    # cached (:1212, then :1216): keep that order
                    exif_bad, exif_data = (exif_data, e), None
        except Exception as e:
            if self.strict_errors:
                raise
            print(f"Guided feature extraction failed: {e!r}")          # the reference's message (:1238)
            return self.forward(images, exif_data, return_attention)
        images, ws, B = self._prepare(images)
        N, D, T = ws["N"], self.feature_dim, ws["T"]
        gkey = ("mask", attention_guidance.lower(), math.isqrt(N)) if isinstance(attention_guidance, str) else None
        gdev = self._table(*gkey) if gkey else g.to(self._packed["device"]).contiguous()
        cdraw = self._curiosity_draw(B) if self.faithful_curiosity else None     # :1185, before the Linear of :1420
        if not self.use_iterative:
            temp_projection = (ws["pw"], ws["pb"])            # unused: the stream's own projection is applied
        if temp_projection is None:
            lin = nn.Linear(self.feature_dim, 64)
            temp_projection = (lin.weight.detach(), lin.bias.detach())
        ws["pw"].copy_(temp_projection[0].reshape(64, D), non_blocking=True)     # static buffers: the captured
        ws["pb"].copy_(temp_projection[1].reshape(64), non_blocking=True)        # sequence reads these addresses
        pw, pb = ws["pw"], ws["pb"]

        def guided_focal(col):
            st = stream_ptr()
            call("cogaim_guided_attention", ptr(ws["attn"]), ptr(gdev), ptr(ws["guided"]), B, N, 0.7, 0.05, st)
            call("cogaim_weighted_tokens", ws["tokens"].data_ptr() + D * 4, T * D, None, None, ptr(ws["guided"]),
                 ptr(ws["wt"]), ptr(ws["wtpart"]), B, N, D, st)
            if self.use_iterative:
                self._linear(ptr(ws["wt"]), D, None, ws["cat"].data_ptr() + col * 4, 192, B, W=pw, bias=pb)
            else:       # a bare FocalStream has its own projection (:1417-1418): no random Linear, deterministic depth
                self._linear(ptr(ws["wt"]), D, "focal_stream.projection.0", ptr(ws["t256"]), 256, B, act=1)
                self._linear(ptr(ws["t256"]), 256, "focal_stream.projection.3", ws["cat"].data_ptr() + col * 4, 192, B)

        ncols = self._features(images, exif_data, ws, B, focal_override=guided_focal,
                               graph_key=("guided", gdev.data_ptr()) if gkey else None)
        if cdraw is not None:
            self._curiosity_run(ws, B, cdraw)
        self._check_fallback_flags(ws)
        attn = ws["guided"].clone()
        self._last_attention_weights = attn                # :1212
        if exif_bad is not None:
            print(f"Guided feature extraction failed: {exif_bad[1]!r}")
            return self.forward(images, exif_bad[0], return_attention)       # -> the all-zero backup features
        if ncols != 192:
            # the reference's fusion Linear(192,192) rejects the short concat and its except
            # branch re-runs the standard forward (:1226, :1237-1240); the guided attention stays cached
            out = self.forward(images, exif_data, return_attention)
            return out
        depth, conf = self._outputs(ws)
        if return_attention:
            return depth, conf, attn
        return depth, conf

    def _guided_sweep(self, images, exif_data, guidances, return_attention, temp_projection):
        """K instructions over one backbone + focal pass (reference src/model.py:1242-1426 applied K times to
        the same ``base_attention``).  Bit-identical to K single calls."""
        K = len(guidances)
        if K == 0:
            raise ValueError("empty instruction list")
        if not self.use_focal or any(g is None for g in guidances) or (images.dim() == 4 and images.shape[0] == 0):
            outs = [self.forward_with_guidance(images, exif_data, g, return_attention, temp_projection=tp)
                    for g, tp in zip(guidances, self._projection_list(temp_projection, K, draw=False))]
            return tuple(torch.stack([o[i] for o in outs]) if outs[0][i] is not None else None
                         for i in range(len(outs[0])))
        images, ws, B = self._prepare(images)
        N, D, T = ws["N"], self.feature_dim, ws["T"]
        dev = self._packed["device"]
        names = tuple(g.lower() if isinstance(g, str) else None for g in guidances)
        mkey = ("masks", names, N) if all(n is not None for n in names) else None
        masks = self._tab.get(mkey) if mkey else None
        if masks is None:
            rows = []
            for g in guidances:
                v = tables.guidance_vector(g, N)
                if v.numel() != N:
                    raise ValueError("attention_guidance cannot be resized to the patch grid")
                rows.append(v.reshape(N).float())
            masks = torch.stack(rows).to(dev).contiguous()
            if mkey:
                self._tab[mkey] = masks
        sw = ws.setdefault("sweep", {}).get(K)
        if sw is None:
            f32 = dict(device=dev, dtype=torch.float32)
            nchunk = (N + call("cogaim_weighted_tokens_rows") - 1) // call("cogaim_weighted_tokens_rows")
            sw = {"guided": torch.empty(K, B, N, **f32), "wt": torch.empty(K * B, D, **f32),
                  "wtpart": torch.empty(K * B * nchunk * D, **f32), "cat": torch.empty(K, B, 192, **f32),
                  "fused": torch.empty(K * B, 192, **f32), "out": torch.empty(2, K * B, **f32),
                  "pw": torch.empty(K, 64, D, **f32), "pb": torch.empty(K, 64, **f32)}
            ws["sweep"][K] = sw
        given = self._projection_list(temp_projection, K, draw=False)
        cdraws, pairs = [], []
        for k in range(K):                 # the random stream of K single calls: curiosity draws (:1185), then the Linear
            if self.faithful_curiosity:
                cdraws.append(self._curiosity_draw(B))
            if not self.use_iterative:
                pairs.append(None)                          # the stream's own projection: nothing drawn
            elif given[k] is None:
                lin = nn.Linear(self.feature_dim, 64)
                pairs.append((lin.weight.detach(), lin.bias.detach()))
            else:
                pairs.append(given[k])
        for k, pair in enumerate(pairs):
            if pair is not None:
                sw["pw"][k].copy_(pair[0].reshape(64, D), non_blocking=True)     # static buffers: the captured sequence
                sw["pb"][k].copy_(pair[1].reshape(64), non_blocking=True)        # reads these addresses
        col_focal = []

        def tail():
            st, f = stream_ptr(), self._packed["f32"]
            col = col_focal[0]
            call("cogaim_guided_attention_multi", ptr(ws["attn"]), ptr(masks), ptr(sw["guided"]), B, N, K, 0.7, 0.05, st)
            call("cogaim_weighted_tokens_multi", ws["tokens"].data_ptr() + D * 4, T * D, ptr(sw["guided"]), ptr(sw["wt"]),
                 ptr(sw["wtpart"]), B, N, D, K, st)
            sw["cat"].copy_(ws["cat"].unsqueeze(0).expand(K, B, 192))      # ambient / EXIF columns of every image
            for k in range(K):
                wt_k, cat_k = sw["wt"].data_ptr() + k * B * D * 4, sw["cat"].data_ptr() + (k * B * 192 + col) * 4
                if self.use_iterative:
                    self._linear(wt_k, D, None, cat_k, 192, B, W=sw["pw"][k], bias=sw["pb"][k])
                else:
                    self._linear(wt_k, D, "focal_stream.projection.0", ptr(ws["t256"]), 256, B, act=1)
                    self._linear(ptr(ws["t256"]), 256, "focal_stream.projection.3", cat_k, 192, B)
            out = sw["out"]
            call("cogaim_fusion_heads", ptr(sw["cat"]), 192, 192, ptr(f["fusion.0.weight"]), ptr(f["fusion.0.bias"]), 192,
                 ptr(f["decision_head.0.weight"]), ptr(f["decision_head.0.bias"]),
                 ptr(f["confidence_head.0.weight"]), ptr(f["confidence_head.0.bias"]),
                 ptr(f["confidence_head.2.weight"]), ptr(f["confidence_head.2.bias"]),
                 ptr(sw["fused"]), out.data_ptr(), out.data_ptr() + K * B * 4, K * B, st)

        ncols = self._features(images, exif_data, ws, B, focal_override=col_focal.append, heads=False,
                               graph_key=("sweep", masks.data_ptr(), K) if mkey else None, tail=tail)
        for dr in cdraws:
            self._curiosity_run(ws, B, dr)
        self._check_fallback_flags(ws)
        attn = sw["guided"].clone()
        self._last_attention_weights = attn[K - 1]            # what the last of K single calls leaves (:1212)
        if ncols != 192:
            # every single call would fall back to forward() (:1226, :1237-1240) with the guided map kept cached
            last = self._last_attention_weights
            out = self.forward(images, exif_data, False)
            self._last_attention_weights = last
            depth, conf = (o.unsqueeze(0).expand(K, B, 1).clone() for o in out)
        else:
            o = sw["out"].clone()
            depth, conf = o[0].reshape(K, B, 1), o[1].reshape(K, B, 1)
        if return_attention:
            return depth, conf, attn
        return depth, conf

    def _projection_list(self, temp_projection, K, draw):
        """K (weight, bias) pairs for the per-call Linear(768,64) of src/model.py:1417-1422: the caller's pair for all,
        the caller's K pairs, or (``draw``) K fresh nn.Linear initialisations in call order."""
        if temp_projection is None:
            if not draw:
                return [None] * K
            pairs = []
            for _ in range(K):
                lin = nn.Linear(self.feature_dim, 64)
                pairs.append((lin.weight.detach(), lin.bias.detach()))
            return pairs
        if isinstance(temp_projection, (list,)) or (isinstance(temp_projection, tuple) and len(temp_projection) == K
                                                    and isinstance(temp_projection[0], (tuple, list))):
            if len(temp_projection) != K:
                raise ValueError(f"{len(temp_projection)} projections for {K} instructions")
            return [tuple(t) for t in temp_projection]
        return [tuple(temp_projection)] * K

    # training-only helpers of the reference that callers may probe for
    def compute_curiosity_loss(self, *a, **k):
        raise NotImplementedError("training is out of scope for this build")

    def get_exploration_stats(self):
        """Statistics of the positive entries of the exploration ring buffer (reference src/model.py:775-793, :1456-1460).
        The buffer is only written in the opt-in ``faithful_curiosity`` mode (or by a loaded checkpoint)."""
        hist = self.curiosity_module.exploration_history.detach().float().cpu()
        valid = hist[hist > 0]
        if valid.numel() == 0:
            return {"mean": 0.0, "std": 0.0, "max": 0.0, "samples": 0}
        return {"mean": float(valid.mean()), "std": float(valid.std()) if valid.numel() > 1 else 0.0,
                "max": float(valid.max()), "min": float(valid.min()), "samples": int(valid.numel())}


def create_model(config, camera_info=None, device=None):
    """Factory with the reference's signature (src/model.py:1534-1586), including the
    optional top-level ``config['load_checkpoint']`` loader that skips head and
    cognitive-module weights (:1549-1577)."""
    model = CognitiveAimModel(config, camera_info)
    if device is not None:
        model = model.to(device)
    if config.get("load_checkpoint"):
        path = config["load_checkpoint"]
        try:
            state = torch.load(path, map_location="cpu", weights_only=True)
            skip = ("decision_head.", "confidence_head.", "curiosity_module.", "global_aligner.",
                    "ambient_stream.", "focal_stream.", "exif_prior.", "fusion.")
            model.load_state_dict({k: v for k, v in state.items() if not k.startswith(skip)}, strict=False)
            if device is not None:
                model = model.to(device)
        except Exception as e:  # same outward behaviour as the reference: warn and keep the init
            print(f"Warning: Unable to load pretrained weights {path}: {e}")
    return model.eval()
