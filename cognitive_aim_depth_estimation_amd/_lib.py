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
    "cogaim_set_gemm_algo": [I],
    "cogaim_im2col": [P, P, I, I, I, I, I, I, P],
    "cogaim_gemm_patch_embed": [P, P, P, P, P, I, I, I, I, I, P, LL, P],
    "cogaim_cls_row": [P, P, P, I, I, I, P],
    "cogaim_layernorm": [P, P, P, P, I, I, F, I, P],
    "cogaim_gemm_bias_half": [P, P, P, P, I, I, I, I, I, I, I, F, I, P, LL, P, P],
    "cogaim_mhsa_qkv_fwd": [P, P, I, I, I, I, I, P],
    "cogaim_gemm_residual_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P, LL, P],
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
    "cogaim_weighted_tokens_static": [P, LL, P, P, P, P, I, I, I, P],
    "cogaim_guided_attention": [P, P, P, I, I, F, F, P],
    "cogaim_guided_attention_multi": [P, P, P, I, I, I, F, F, P],
    "cogaim_weighted_tokens_multi": [P, LL, P, P, P, I, I, I, I, P],
    "cogaim_linear_f32": [P, I, P, P, P, P, I, I, I, I, I, P],
    "cogaim_exif_prep": [P, P, P, P, P, I, I, P, I, P, I, P, P],
    "cogaim_heads": [P, I, P, P, P, P, P, P, P, P, I, P],
    "cogaim_heatmap_postprocess": [P, P, I, I, I, P],
    "cogaim_curiosity_sample": [P, P, P, P, P, LL, P, P, I, I, I, P],
    "cogaim_curiosity_finish": [P, P, LL, P, P, P, P, P, P, P, P, P, I, I, I, P],
}

_NO_STATUS = {"cogaim_version", "cogaim_gemm_workspace_bytes", "cogaim_focal_colsum_rows", "cogaim_weighted_tokens_rows",
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
