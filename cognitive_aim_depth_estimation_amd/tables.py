"""Weight-free lookup tables of the focal stream and the guided path.

They depend only on the patch grid, are built once per grid on the host in
fp32 and cached on the device; the kernels consume them as plain arrays.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Union

import torch
import torch.nn.functional as F

_FOCUS = {
    # name -> (row numerator, col numerator) in quarters of the grid
    "left": (2, 1), "right": (2, 3), "top": (1, 2), "bottom": (3, 2),
    "top-left": (1, 1), "topleft": (1, 1), "top-right": (1, 3), "topright": (1, 3),
    "bottom-left": (3, 1), "bottomleft": (3, 1), "bottom-right": (3, 3), "bottomright": (3, 3),
}


@lru_cache(maxsize=16)
def position_table(num_patches: int, dim: int) -> torch.Tensor:
    """Sinusoidal table added to the patch tokens in every FocalStream call
    (reference src/model.py:140-177).  Square grids: channels [0, dim/2) encode
    the row and [dim/2, dim) the column, interleaved sin/cos with frequencies
    exp(-2k ln(1e4) / (dim/2)).  Non-square counts: classic 1-D table."""
    g = math.isqrt(num_patches)
    pe = torch.zeros(num_patches, dim, dtype=torch.float32)
    if g * g == num_patches:
        if dim < 4 or dim % 4:
            return pe       # the reference skips the encoding when the slices cannot be filled
        half = dim // 2
        freq = torch.exp(torch.arange(0, half, 2, dtype=torch.float32) * -(math.log(10000.0) / half))
        n = torch.arange(num_patches)
        row = (n // g).to(torch.float32)[:, None] * freq[None, :]
        col = (n % g).to(torch.float32)[:, None] * freq[None, :]
        pe[:, 0:half:2] = torch.sin(row)
        pe[:, 1:half:2] = torch.cos(row)
        pe[:, half::2] = torch.sin(col)
        pe[:, half + 1::2] = torch.cos(col)
    else:
        freq = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
        ang = torch.arange(num_patches, dtype=torch.float32)[:, None] * freq[None, :]
        pe[:, 0::2] = torch.sin(ang)
        if dim > 1:
            pe[:, 1::2] = torch.cos(ang)[:, :dim // 2]
    return pe


@lru_cache(maxsize=16)
def center_bias_table(num_patches: int, strength: float = 0.3) -> torch.Tensor:
    """0.3 * Gaussian bump centred on patch (G//2, G//2) with sigma G/6
    (reference src/model.py:208-231; 1-D variant for non-square counts)."""
    g = math.isqrt(num_patches)
    if g * g != num_patches:
        d = (torch.arange(num_patches, dtype=torch.float32) - (num_patches // 2)).abs()
        sigma = num_patches / 12
        return torch.exp(-d ** 2 / (2 * sigma ** 2)) * strength
    ys, xs = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
    c = g // 2
    d = torch.sqrt((xs - c).float() ** 2 + (ys - c).float() ** 2)
    sigma = g / 6
    return (torch.exp(-d ** 2 / (2 * sigma ** 2)) * strength).flatten()


@lru_cache(maxsize=64)
def instruction_mask(instruction: str, g: int) -> torch.Tensor:
    """G*G spatial prior for a named instruction (reference src/model.py:1262-1379):
    'center' -> disc of radius max(1, G//4) at the grid centre weighted 3.0, ring
    out to twice that 1.5; the eight directional names -> disc radius max(1, G//6)
    at the quarter points weighted 5.0, ring 2.0; anything else -> all ones."""
    name = instruction.lower()
    if name == "center":
        fy, fx, r, hi, mid = g // 2, g // 2, max(1, g // 4), 3.0, 1.5
    elif name in _FOCUS:
        qy, qx = _FOCUS[name]
        fy, fx, r, hi, mid = g * qy // 4, g * qx // 4, max(1, g // 6), 5.0, 2.0
    else:
        return torch.ones(g * g, dtype=torch.float32)
    ys, xs = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
    d2 = (ys - fy) ** 2 + (xs - fx) ** 2          # exact integers: d <= r  <=>  d2 <= r*r
    mask = torch.ones(g, g, dtype=torch.float32)
    mask[d2 <= (2 * r) ** 2] = mid
    mask[d2 <= r * r] = hi
    return mask.flatten()


def guidance_vector(guidance: Union[str, torch.Tensor], num_patches: int) -> torch.Tensor:
    """String -> mask; tensor of another (square) length -> bilinear resize to the
    patch grid, align_corners=False (reference src/model.py:1385-1398)."""
    if isinstance(guidance, str):
        return instruction_mask(guidance, math.isqrt(num_patches))
    g = guidance.detach().to("cpu", torch.float32).flatten()
    if g.numel() != num_patches:
        gs, ts = math.isqrt(g.numel()), math.isqrt(num_patches)
        g = F.interpolate(g.view(1, 1, gs, gs), size=(ts, ts), mode="bilinear",
                          align_corners=False).flatten()
    return g


def interpolate_position_embeddings(pos: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """DINOv2 position table for a gh x gw grid, [1+gh*gw, D] (hf:57-95): the
    native table when the grid is the native square one, else bicubic
    (align_corners=False, fp32) resampling of the patch part."""
    pos = pos.detach().to(torch.float32)
    n = pos.shape[1] - 1
    if gh * gw == n and gh == gw:
        return pos[0].contiguous()
    s = math.isqrt(n)
    d = pos.shape[-1]
    grid = pos[:, 1:].reshape(1, s, s, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).reshape(gh * gw, d)
    return torch.cat((pos[0, :1], grid), dim=0).contiguous()


# --------------------------------------------------------------------------
# image preprocessing tables (reference demo.py:162-166: Resize -> ToTensor -> Normalize)
# --------------------------------------------------------------------------
RESAMPLE_PRECISION_BITS = 22      # Pillow's 8-bit resampler: 32 - 8 - 2 fractional bits per weight


def resample_coefficients(in_size: int, out_size: int):
    """Per output index of one image axis: (first source index, tap count) and the integer bilinear weights
    Pillow's ``Image.resize(..., BILINEAR)`` uses (triangle filter whose support grows with the shrink factor,
    weights normalised in double precision and rounded to 22 fractional bits).  Returns
    ``bounds int32 [out_size, 2]`` and ``coefs int32 [out_size, ksize]`` (unused taps are zero)."""
    import numpy as np
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)              # C (int) cast: truncation
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w = np.where(taps < xmax[:, None], w, 0.0)
    # the reference sums the weights left to right; cumulative summation keeps that order exactly
    ww = np.cumsum(w, axis=1)[:, -1:]
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    q = w * float(1 << RESAMPLE_PRECISION_BITS)
    coefs = np.where(w < 0, (-0.5 + q), (0.5 + q)).astype(np.int64).astype(np.int32)   # truncation, as (int)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, coefs


def normalize_lut(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """float32 [3, 256]: ToTensor's v / 255 followed by Normalize's (x - mean) / std for every byte value,
    evaluated with the same float32 operations as the reference's transforms."""
    v = torch.arange(256, dtype=torch.float32).div(255)
    return torch.stack([v.sub(torch.tensor(mean[c])).div(torch.tensor(std[c])) for c in range(3)])
