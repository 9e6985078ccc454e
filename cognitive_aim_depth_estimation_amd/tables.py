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
