"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image preprocessing in front of the forward pass.

The reference (demo.py:162-166, applied at demo.py:319) runs torchvision's
``Resize(size) -> ToTensor() -> Normalize(mean, std)`` on a PIL RGB image.  For PIL input ``Resize`` is
``Image.resize((w, h), BILINEAR)``, i.e. Pillow's ImagingResample (src/libImaging/Resample.c of the
installed Pillow): per axis a table of bilinear (triangle) weights whose support widens with the shrink
factor, quantised to 22 fractional bits, a horizontal pass into an 8-bit intermediate, then a vertical pass,
each rounding with +2^21 and clipping to 0..255.  ``ToTensor`` is uint8 -> float32 / 255 in CHW order and
``Normalize`` is (x - mean) / std in float32.

Plain loops over output pixels on purpose (small cases only); pinned bit-for-bit against Pillow itself in
tests/test_preprocess_oracle.py.  Only tests may import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _triangle(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def coefficients(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the full axis.
    Returns (bounds[out][2] = (first source index, tap count), coefs[out][ksize] as Python ints)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, coefs = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        k = k + [0.0] * (ksize - xmax)
        ki = [int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in k]
        bounds.append((xmin, xmax))
        coefs.append(ki)
    return bounds, coefs


def _clip8(v):
    v >>= PRECISION_BITS
    return 0 if v < 0 else (255 if v > 255 else v)


def resize_bilinear_u8(img, out_h, out_w):
    """img: uint8 [h, w, 3] -> uint8 [out_h, out_w, 3]; horizontal pass first, 8-bit intermediate."""
    h, w, _ = img.shape
    xb, xk = coefficients(w, out_w)
    yb, yk = coefficients(h, out_h)
    tmp = np.zeros((h, out_w, 3), np.uint8)
    for y in range(h):
        for xx in range(out_w):
            x0, n = xb[xx]
            for c in range(3):
                ss = 1 << (PRECISION_BITS - 1)
                for t in range(n):
                    ss += int(img[y, x0 + t, c]) * xk[xx][t]
                tmp[y, xx, c] = _clip8(ss)
    out = np.zeros((out_h, out_w, 3), np.uint8)
    for yy in range(out_h):
        y0, n = yb[yy]
        for x in range(out_w):
            for c in range(3):
                ss = 1 << (PRECISION_BITS - 1)
                for t in range(n):
                    ss += int(tmp[y0 + t, x, c]) * yk[yy][t]
                out[yy, x, c] = _clip8(ss)
    return out


def to_tensor_normalize(img_u8):
    """uint8 [h, w, 3] -> float32 [3, h, w]: / 255, then (x - mean) / std, all in float32."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    mean = np.asarray(MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(STD, np.float32).reshape(3, 1, 1)
    return (x - mean) / std


def preprocess(img_u8, out_h, out_w):
    return to_tensor_normalize(resize_bilinear_u8(img_u8, out_h, out_w))
