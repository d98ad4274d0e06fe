"""Restatement of Pillow's 8-bit bilinear resampler (oracle/pil_resize.py) and the library's host-side
coefficient tables (vamd_resize_coeffs) against Pillow itself -- the third-party code torchvision's Resize
runs for the reference's data module (reference dataloader.py:32-36).  CPU only."""
import ctypes as C

import numpy as np
import pytest
from PIL import Image

from oracle import pil_resize as PR

SIZES = [(37, 53, 16, 16), (300, 200, 224, 224), (224, 224, 224, 224), (100, 224, 224, 224), (512, 640, 224, 224),
         (16, 16, 32, 48), (448, 7, 64, 64), (9, 500, 64, 64), (225, 223, 224, 224), (2, 3, 5, 4), (1024, 1024, 1024, 512)]


@pytest.mark.parametrize("shape", SIZES, ids=lambda s: "%dx%d-%dx%d" % s)
def test_restatement_equals_pillow(shape):
    """Aspect ratios up to 64:1.  (Pillow 12.2 was seen to change the order of its two passes on images more than 100
    times taller than wide when it shrinks them vertically, e.g. 701 x 7 -> 64 x 64; the restatement and the GPU
    pipeline do not follow it there, and the data module keeps such images on the Pillow path.)"""
    H, W, Ho, Wo = shape
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img[: H // 2, : W // 3] = 255                        # saturated and flat regions: clipping and exact weights
    img[H // 2:, : W // 4] = 0
    want = np.asarray(Image.fromarray(img, "RGB").resize((Wo, Ho), Image.BILINEAR))
    got = PR.resize_u8(img, Ho, Wo)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("pair", [(53, 16), (200, 224), (640, 224), (16, 48), (7, 64), (223, 224), (1024, 224), (3, 4)],
                         ids=lambda p: "%d-%d" % p)
def test_library_coefficient_tables_equal_restatement(pair):
    from vision_amd._lib import call, lib
    n_in, n_out = pair
    bounds, coef = PR.precompute_coeffs(n_in, n_out)
    ks = lib.vamd_resize_ksize(n_in, n_out)
    assert ks == coef.shape[1]
    b = np.full((n_out, 2), -1, dtype=np.int32)
    c = np.full((n_out, ks), -1, dtype=np.int32)
    call("vamd_resize_coeffs", n_in, n_out, b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), ks)
    assert np.array_equal(b, bounds) and np.array_equal(c, coef)


def test_to_tensor_division():
    import torch
    v = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    assert np.array_equal(PR.to_tensor(v), (torch.from_numpy(v).permute(2, 0, 1).float() / 255).numpy())
