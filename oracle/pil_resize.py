"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the per-sample transform of the reference data module
(reference dataloader.py:32-36): torchvision `Resize((p, p))` on a PIL image followed by `ToTensor()`.

The arithmetic lives in a third-party dependency, not in the reference: torchvision's Resize calls
`PIL.Image.resize(size, BILINEAR)` (Pillow; 12.2.0 in this image), whose resampler is
src/libImaging/Resample.c: `precompute_coeffs` (double precision windows and triangle weights),
`normalize_coeffs_8bpc` (22 fractional bits), `ImagingResampleHorizontal_8bpc` then
`ImagingResampleVertical_8bpc` (the first pass is rounded to 8 bits before the second).  This file restates
those functions; tests pin it on Pillow itself, which IS importable here and on the GPU box.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """-> (bounds int32 [out][2] = (first source index, count), coef int32 [out][ksize])"""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k, ww = [], 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        k = [v / ww if ww != 0.0 else v for v in k]
        bounds[xx] = (xmin, xmax)
        for x, v in enumerate(k):
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return bounds, coef


def _pass(img, bounds, coef, axis):
    """img uint8 [H][W][C]; resample `axis` (0 rows, 1 columns)."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for i, (lo, n) in enumerate(bounds):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j in range(n):
            acc += src[lo + j] * int(coef[i, j])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img, out_h, out_w):
    """Image.resize((out_w, out_h), BILINEAR) of an RGB uint8 [H][W][3] array: horizontal pass, then vertical."""
    H, W, _ = img.shape
    if W != out_w:
        img = _pass(img, *precompute_coeffs(W, out_w), axis=1)
    if H != out_h:
        img = _pass(img, *precompute_coeffs(H, out_h), axis=0)
    return img


def to_tensor(img):
    """transforms.ToTensor on an 8-bit image: [H][W][C] uint8 -> [C][H][W] float32 = value / 255 (fp32 division)."""
    return (np.moveaxis(img, 2, 0).astype(np.float32) / np.float32(255.0)).astype(np.float32)
