"""ctypes binding of libcogaim.so (the C ABI declared in include/cogaim.h).

The library is mandatory: there is no eager / CPU fallback.  ``lib()`` raises
if it has not been built (``python -m cognitive_aim_depth_estimation_amd.build``).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COGAIM_LIB") or os.path.join(HERE, "libcogaim.so")     # override: A/B runs in tools/

P, I, F, LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argument types (all return int); mirrors include/cogaim.h
SIGNATURES = {
    "cogaim_version": [],
    "cogaim_preprocess_rgb8": [P, I, I, LL, I, P, P, I, P, P, I, P, P, P, I, I, P],
    "cogaim_preprocess_rgb8_batch": [P, I, I, P, I, I, P],
    "cogaim_gemm_workspace_bytes": [],
    "cogaim_im2col": [P, P, I, I, I, I, I, I, P],
    "cogaim_gemm_patch_embed": [P, P, P, P, P, I, I, I, I, I, P, LL, P],
    "cogaim_cls_row": [P, P, P, I, I, I, P],
    "cogaim_layernorm": [P, P, P, P, I, I, F, I, P],
    "cogaim_gemm_bias_half": [P, P, P, P, I, I, I, I, I, I, I, F, I, P, LL, P, P],
    "cogaim_mhsa_qkv_fwd": [P, P, I, I, I, I, I, P],
    "cogaim_gemm_residual_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P, LL, P],
    "cogaim_gemm_residual_layernorm": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, P],
    "cogaim_gemm_bias_gelu_half": [P, P, P, P, I, I, I, I, I, I, I, P, LL, P, P],
    "cogaim_focal_prep": [P, LL, P, P, P, I, I, I, I, P],
    "cogaim_gemm_batched_f32": [P, P, P, I, I, I, I, I, I, LL, LL, LL, I, I, P],
    "cogaim_focal_colsum_rows": [],
    "cogaim_focal_softmax_stats_colsum": [P, I, LL, P, P, P, I, I, P],
    "cogaim_focal_attention_candidates": [P, I, P, P, I, LL, P, P, P, P, I, I, P],
    "cogaim_focal_attention_vector": [P, P, I, P, P, P, F, I, I, P, I, P],
    "cogaim_focal_weighted_colsum": [P, I, LL, P, P, P, P, I, I, P],
    "cogaim_focal_reduce_u": [P, P, P, I, I, I, P],
    "cogaim_focal_rowstats_parts": [I],
    "cogaim_focal_scores_colsum_blocks": [I],
    "cogaim_focal_scores_rowstats": [P, P, P, P, I, I, I, LL, I, I, P],
    "cogaim_focal_rowstats_combine": [P, P, I, P, P, I, I, P],
    "cogaim_focal_scores_colsum": [P, P, P, P, P, P, I, I, I, LL, I, I, P],
    "cogaim_weighted_tokens_rows": [],
    "cogaim_weighted_tokens": [P, LL, P, P, P, P, P, I, I, I, P],
    "cogaim_weighted_tokens_parts": [P, LL, P, P, P, I, P, P, P, I, I, I, P],
    "cogaim_weighted_tokens_static": [P, LL, P, P, P, P, I, I, I, P],
    "cogaim_guided_attention": [P, P, P, I, I, F, F, P],
    "cogaim_guided_attention_multi": [P, P, P, I, I, I, F, F, P],
    "cogaim_weighted_tokens_multi": [P, LL, P, P, P, I, I, I, I, P],
    "cogaim_linear_f32": [P, I, P, P, P, P, I, P, I, I, I, I, I, P],
    "cogaim_exif_prep": [P, P, P, P, P, I, I, P, I, P, I, P, P],
    "cogaim_heads": [P, I, P, P, P, P, P, P, P, P, I, P],
    "cogaim_mlp_chain": [P, I, P],
    "cogaim_fusion_heads": [P, I, I, P, P, I, P, P, P, P, P, P, P, P, P, I, P],
    "cogaim_export_outputs": [P, P, P, I, P, I, P, P, P, P, P, P, I, I, P],
    "cogaim_host_pointer_ok": [P],
    "cogaim_allgather_f32": [P, P, LL, P, P],
    "cogaim_heatmap_postprocess": [P, P, I, I, I, P],
    "cogaim_curiosity_sample": [P, P, P, P, P, LL, P, P, I, I, I, P],
    "cogaim_curiosity_finish": [P, P, LL, P, P, P, P, P, P, P, P, P, I, I, I, P],
}



class MlpLayer(C.Structure):
    """cogaim_mlp_layer of include/cogaim.h."""
    _fields_ = [("W", P), ("b", P), ("b2", P), ("K", I), ("O", I), ("act", I), ("in_off", I), ("out_off", I), ("reserved", I)]


class MlpChainDesc(C.Structure):
    """cogaim_mlp_chain_desc of include/cogaim.h."""
    _fields_ = [("layer", MlpLayer * 8), ("n_layers", I), ("in0_len", I), ("in0_off", I), ("in0_parts", I),
                ("in0", P), ("in0_ld", LL), ("in0_part_stride", LL),
                ("in1", P), ("in1_ld", LL), ("in1_len", I), ("in1_off", I),
                ("scale", P), ("out", P), ("out_ld", LL), ("out_len", I), ("out_off", I)]


def mlp_chain(layers, in0, in0_ld, in0_len, out, out_ld, out_len, out_off, B, in0_off=0, in0_parts=1, in0_part_stride=0,
              in1=None, in1_ld=0, in1_len=0, in1_off=0, scale=None):
    """cogaim_mlp_chain: `layers` = [(W ptr, b ptr, b2 ptr, K, O, act, in_off, out_off), ...] (at most eight)."""
    d = MlpChainDesc()
    for i, (W, b, b2, K, O, act, io, oo) in enumerate(layers):
        d.layer[i] = MlpLayer(W, b, b2, K, O, act, io, oo, 0)
    d.n_layers, d.in0_len, d.in0_off, d.in0_parts = len(layers), in0_len, in0_off, in0_parts
    d.in0, d.in0_ld, d.in0_part_stride = in0, in0_ld, in0_part_stride
    d.in1, d.in1_ld, d.in1_len, d.in1_off = in1, in1_ld, in1_len, in1_off
    d.scale, d.out, d.out_ld, d.out_len, d.out_off = scale, out, out_ld, out_len, out_off
    return call("cogaim_mlp_chain", C.byref(d), B, stream_ptr())


# exported by -DCOGAIM_TOOLS builds only (tools/wt_build.sh): bound when the loaded library has them
OPTIONAL = {"cogaim_set_gemm_algo": [I]}

_NO_STATUS = {"cogaim_version", "cogaim_host_pointer_ok", "cogaim_gemm_workspace_bytes", "cogaim_focal_colsum_rows", "cogaim_weighted_tokens_rows",
              "cogaim_focal_rowstats_parts", "cogaim_focal_scores_colsum_blocks"}
_lib = None


class CogaimError(RuntimeError):
    pass


def lib() -> C.PyDLL:
    """The library handle.  Loaded with ctypes.PyDLL: every entry point only enqueues work on a stream and returns in
    microseconds, and a call that dropped the interpreter lock would have to win it back from whatever other thread
    holds it (decode workers of the batch loop), which costs far more than the call."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CogaimError(
                f"{LIB_PATH} is missing: build it with "
                "`python -m cognitive_aim_depth_estimation_amd.build` (hipcc, gfx950). "
                "This package has no CPU or eager fallback.")
        handle = C.PyDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = LL if name == "cogaim_gemm_workspace_bytes" else I
        for name, args in OPTIONAL.items():
            if hasattr(handle, name):
                fn = getattr(handle, name)
                fn.argtypes, fn.restype = args, I
        _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


# When set to a list, every launch is bracketed by events on the launch stream and
# (name, args, start_event, end_event) is appended: per-kernel timing for bench.py.
PROFILE = None


def call(name: str, *args):
    if PROFILE is not None and name not in _NO_STATUS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib(), name)(*args)
        e1.record()
        PROFILE.append((name, args, e0, e1))
    else:
        rc = getattr(lib(), name)(*args)
    if name not in _NO_STATUS and rc != 0:
        raise CogaimError(f"{name} failed with code {rc}")
    return rc
