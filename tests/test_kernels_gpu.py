"""Per-kernel parity of libvision_amd.so (through the C ABI) against fp64 torch-CPU evaluations of the
same operator on the same (dtype-rounded) inputs.  Run on the MI355X box: pytest -m gpu."""
import ctypes as C
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from vision_amd import _lib                      # noqa: E402
from vision_amd._lib import ConvDesc, call, lib   # noqa: E402

DEV = torch.device("cuda:0")
DT = {"f32": (torch.float32, _lib.F32), "bf16": (torch.bfloat16, _lib.BF16)}
# relative Frobenius-norm error allowed for outputs stored in the compute dtype / in fp32
TOL_OUT = {"f32": 2e-5, "bf16": 4e-3}
TOL_F32OUT = {"f32": 2e-5, "bf16": 2e-4}


def S():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []


def P(t):
    """Device pointer of t; t is kept alive until the end of the test (a temporary passed by
    pointer would otherwise be freed, and its block reused by the next temporary, before the
    asynchronous kernel reads it)."""
    if t is None:
        return None
    _KEEP.append(t)
    return t.data_ptr()


@pytest.fixture(autouse=True)
def _release_temporaries():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(shape, tdt, gen, scale=1.0):
    """Random tensor rounded to the compute dtype; returns (device tensor in tdt, fp64 CPU copy)."""
    x = (torch.randn(shape, generator=gen) * scale).to(tdt)
    return x.to(DEV), x.double()


def nhwc(x_nchw64):
    return x_nchw64.permute(0, 2, 3, 1).contiguous()


def pack_weights(w64, tdt, dt):
    """fp32 OIHW -> packed forward / dgrad copies through vamd_repack_weight."""
    O, I, k, _ = w64.shape
    w32 = w64.float().to(DEV).contiguous()
    wf = torch.empty((O, k, k, I), dtype=tdt, device=DEV)
    wd = torch.empty((I, k, k, O), dtype=tdt, device=DEV)
    call("vamd_repack_weight", P(w32), P(wf), P(wd), O, I, k, dt, S())
    return wf, wd


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, up
    (2, 12, 12, 64, 64, 3, 1, 0),
    (3, 9, 7, 32, 64, 3, 1, 0),
    (2, 10, 10, 64, 32, 3, 1, 0),
    (1, 13, 11, 128, 192, 3, 1, 0),
    (1, 20, 26, 256, 256, 3, 1, 0),
    (2, 16, 32, 64, 64, 3, 1, 0),
    (2, 19, 37, 64, 64, 3, 1, 0),      # ragged 8 x 32 patches of the resident-filter kernel (when forced)
    (3, 33, 20, 64, 64, 3, 1, 0),      # ragged 16 x 16 patches
    (2, 20, 24, 64, 128, 3, 1, 0),     # two column tiles of the resident-filter kernel (forward)
    (2, 12, 20, 128, 64, 3, 1, 0),     # ... (input gradient)
    (1, 24, 48, 128, 128, 3, 1, 0),
    (1, 8, 16, 64, 64, 3, 1, 1),
    (1, 32, 64, 64, 128, 3, 2, 0),
    (1, 9, 11, 256, 256, 3, 1, 1),
    (2, 18, 18, 256, 512, 1, 1, 0),
    (1, 18, 22, 256, 256, 3, 2, 0),
    (2, 12, 12, 64, 64, 3, 2, 0),
    (2, 7, 9, 32, 32, 3, 2, 0),
    (2, 6, 5, 64, 64, 3, 1, 1),
    (1, 10, 37, 64, 128, 3, 1, 1),     # upsampling forward on the resident-filter kernel (when forced): two column tiles x four classes
    (2, 9, 9, 64, 128, 1, 1, 0),
    (2, 16, 16, 3, 32, 3, 1, 0),
    (2, 16, 16, 32, 3, 3, 1, 0),
    (2, 8, 8, 3, 3, 3, 1, 0),
    (2, 10, 24, 3, 3, 3, 1, 0),
    (2, 9, 12, 3, 3, 3, 1, 0),
    (3, 5, 40, 3, 3, 1, 1, 0),
    (2, 8, 8, 3, 32, 1, 1, 0),
    (2, 8, 8, 3, 3, 1, 1, 0),
    (3, 37, 29, 3, 64, 3, 1, 0),
    (3, 37, 29, 64, 3, 3, 1, 0),
    (2, 20, 20, 64, 3, 1, 1, 0),
    (2, 20, 20, 3, 64, 1, 1, 0),
    (2, 19, 23, 3, 40, 3, 1, 0),
]


# Shapes with fewer row tiles than the chip has CUs: with a scratch buffer in the descriptor the kernels may split such a
# layer's reduction between workgroups (partial products summed in a fixed order); without one they run whole tiles.
# B, H, W, Cin, Cout, k, stride, up
SPLIT_CASES = [
    (4, 56, 56, 256, 256, 3, 1, 0),       # 49 tiles of 256 x 256
    (2, 112, 112, 128, 128, 3, 1, 0),     # 49 tiles of 512 x 128
    (8, 28, 28, 256, 512, 3, 1, 0),       # two column tiles, 25 row tiles each
    (32, 14, 14, 512, 512, 3, 1, 1),      # upsampling conv: its input gradient is a 4 x 4 stride-2 conv (K = 16 * 512)
    (7, 223, 111, 128, 128, 3, 2, 0),     # stride 2, ragged last tile
]

# Shapes with at least 200 tiles: every workgroup of the wide kernels walks whole tiles, several of them in a row.
WIDE_CASES = [
    (8, 128, 128, 64, 128, 3, 1, 0),      # 512 tiles of 256 x 128
    (4, 128, 128, 128, 256, 3, 1, 0),     # 256 tiles of 256 x 256 (reduction of 1152: the longest this tile serves)
    (17, 223, 111, 128, 128, 3, 2, 0),    # stride 2, 209 tiles, ragged last tile
    (8, 128, 128, 128, 128, 1, 1, 0),     # 1 x 1: two K-steps per tile (fewer than the load ring has stages)
    (8, 128, 128, 64, 128, 1, 1, 0),      # one K-step per tile
    (8, 128, 128, 32, 256, 1, 1, 0),      # half a K-step per tile
]


@pytest.mark.parametrize("case", SPLIT_CASES + WIDE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wide_tiles(case):
    """Forward (with bias, time embedding, residual, skip and fused BatchNorm statistics) and input gradient with the
    descriptor's optional scratch buffer against fp64, and against the same calls without it (whole tiles): the
    two differ in fp32 summation order at most.  Run twice with the buffer: nothing a launch leaves in it may reach the
    next one, and the results are bit-reproducible."""
    B, H, W, Cin, Cout, k, stride, up = case
    tdt, dt = DT["bf16"]
    gen = torch.Generator().manual_seed(sum(case))
    pad = 1 if k == 3 else 0
    x, x64 = rnd((B, Cin, H, W), tdt, gen)
    _, w64 = rnd((Cout, Cin, k, k), tdt, gen, 1.0 / math.sqrt(Cin * k * k))
    wf, wd = pack_weights(w64, tdt, dt)
    bias, temb = torch.randn(Cout, generator=gen), torch.randn(B, Cout, generator=gen)
    x64 = x64.requires_grad_(True)
    y64 = conv_ref(x64, w64, stride, up, k)
    res, res64 = rnd(tuple(y64.shape), tdt, gen)
    res2, res264 = rnd(tuple(y64.shape), tdt, gen)
    dy, dy64 = rnd(tuple(y64.shape), tdt, gen)
    add, add64 = rnd((B, Cin, H, W), tdt, gen)
    want = y64.detach() + bias.double()[None, :, None, None] + temb.double()[:, :, None, None] + res64 + res264
    (gx,) = torch.autograd.grad(y64, [x64], dy64)
    xn, dyn, resn, res2n, addn = nhwc(x), nhwc(dy), nhwc(res), nhwc(res2), nhwc(add)
    Ho, Wo = y64.shape[2], y64.shape[3]
    scratch = torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
    outs = {}
    for name, sc in (("plain", None), ("split", scratch), ("split again", scratch)):
        d = ConvDesc(B, H, W, Cin, Cout, k, stride, pad, up, dt, _lib.ALGO_MFMA)
        if sc is not None:
            d.scratch, d.scratch_bytes = sc.data_ptr(), sc.numel()
        nparts = lib.vamd_conv2d_stats_parts(C.byref(d))
        y = torch.full((B, Ho, Wo, Cout), 7.0, dtype=tdt, device=DEV)
        if up:                                   # the upsampling forward runs as four classes on other kernels
            call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), P(temb.to(DEV)), P(resn), P(res2n), P(y), S())
        else:
            assert nparts > 0
            part = torch.full((nparts, Cout, 2), 7.0, dtype=torch.float32, device=DEV)
            call("vamd_conv2d_fwd_stats", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), P(temb.to(DEV)), P(resn), P(res2n), P(y),
                 P(part), S())
            yd = y.double().cpu().reshape(-1, Cout)
            assert relerr(part.sum(0), torch.stack([yd.sum(0), (yd * yd).sum(0)], 1)) < 1e-5, name
        assert relerr(y.permute(0, 3, 1, 2), want) < TOL_OUT["bf16"], name
        dx = torch.full((B, H, W, Cin), 7.0, dtype=tdt, device=DEV)
        if up and lib.vamd_conv2d_dgrad_up4_supported(C.byref(d)):
            wd4 = torch.empty((Cin, 4, 4, Cout), dtype=tdt, device=DEV)
            call("vamd_repack_weight_up4", P(w64.float().to(DEV).contiguous()), P(wd4), Cout, Cin, dt, S())
            call("vamd_conv2d_dgrad_up4", C.byref(d), P(dyn), P(wd4), P(addn), P(dx), S())
        else:
            nb = lib.vamd_conv2d_dgrad_workspace(C.byref(d))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
            call("vamd_conv2d_dgrad", C.byref(d), P(dyn), P(wd), P(addn), P(dx), P(ws), nb, S())
        assert relerr(dx.permute(0, 3, 1, 2), gx + add64) < TOL_OUT["bf16"], name
        outs[name] = (y, dx)
        # bias-only forward and a plain (not accumulating) input gradient: epilogues without a loaded term
        y0 = torch.full((B, Ho, Wo, Cout), 7.0, dtype=tdt, device=DEV)
        call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), None, None, None, P(y0), S())
        assert relerr(y0.permute(0, 3, 1, 2), y64.detach() + bias.double()[None, :, None, None]) < TOL_OUT["bf16"], name
        if not up:
            dx0 = torch.full((B, H, W, Cin), 7.0, dtype=tdt, device=DEV)
            call("vamd_conv2d_dgrad", C.byref(d), P(dyn), P(wd), None, P(dx0), P(ws), nb, S())
            assert relerr(dx0.permute(0, 3, 1, 2), gx) < TOL_OUT["bf16"], name
    # same inputs, same split: bit-reproducible; against whole tiles: rounding of a different summation order
    assert torch.equal(outs["split"][0], outs["split again"][0]) and torch.equal(outs["split"][1], outs["split again"][1])
    assert relerr(outs["split"][0], outs["plain"][0].double()) < 2e-3
    assert relerr(outs["split"][1], outs["plain"][1].double()) < 2e-3


def conv_ref(x64, w64, stride, up, k):
    if up:
        x64 = F.interpolate(x64, scale_factor=2, mode="nearest")
    return F.conv2d(x64, w64, None, stride=stride, padding=1 if k == 3 else 0)


def mfma_ok(cin, cout, dts):
    piece, kstep = (8, 64) if dts == "bf16" else (4, 32)
    return cin % piece == 0 and 2 * cin >= kstep and cout % piece == 0


@pytest.mark.parametrize("algo", [_lib.ALGO_DIRECT, _lib.ALGO_MFMA, _lib.ALGO_THIN])
@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dts, algo):
    B, H, W, Cin, Cout, k, stride, up = case
    if algo == _lib.ALGO_MFMA and not (mfma_ok(Cin, Cout, dts) and mfma_ok(Cout, Cin, dts)):
        pytest.skip("shape outside the MFMA path")
    if algo == _lib.ALGO_THIN and not (stride == 1 and not up and min(Cin, Cout) <= 4):
        pytest.skip("shape outside the thin-channel kernels")
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    x, x64 = rnd((B, Cin, H, W), tdt, gen)
    w_, w64 = rnd((Cout, Cin, k, k), tdt, gen, 1.0 / math.sqrt(Cin * k * k))
    wf, wd = pack_weights(w64, tdt, dt)
    x64.requires_grad_(True)
    w64.requires_grad_(True)
    y64 = conv_ref(x64, w64, stride, up, k)
    Ho, Wo = y64.shape[2], y64.shape[3]
    bias = torch.randn(Cout, generator=gen)
    temb = torch.randn(B, Cout, generator=gen)
    res, res64 = rnd((B, Cout, Ho, Wo), tdt, gen)
    res2, res264 = rnd((B, Cout, Ho, Wo), tdt, gen)
    full64 = y64 + bias.double()[None, :, None, None] + temb.double()[:, :, None, None] + res64 + res264

    d = ConvDesc(B, H, W, Cin, Cout, k, stride, 1 if k == 3 else 0, up, dt, algo)
    xn = nhwc(x).contiguous()
    y = torch.empty((B, Ho, Wo, Cout), dtype=tdt, device=DEV)
    # plain conv
    call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), None, None, None, None, P(y), S())
    assert relerr(y.permute(0, 3, 1, 2), y64.detach()) < TOL_OUT[dts]
    # fused epilogue
    call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), P(temb.to(DEV)), P(nhwc(res)), P(nhwc(res2)), P(y), S())
    assert relerr(y.permute(0, 3, 1, 2), full64.detach()) < TOL_OUT[dts]

    # statistics fused into the epilogue: column sums of the stored outputs and of their squares
    nparts = lib.vamd_conv2d_stats_parts(C.byref(d))
    assert (nparts > 0) == bool(lib.vamd_conv2d_uses_mfma(C.byref(d), 0))
    if nparts > 0:
        y2 = torch.empty_like(y)
        part = torch.full((nparts, Cout, 2), 7.0, dtype=torch.float32, device=DEV)
        call("vamd_conv2d_fwd_stats", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), P(temb.to(DEV)), P(nhwc(res)),
             P(nhwc(res2)), P(y2), P(part), S())
        assert torch.equal(y2, y)
        yd = y.double().cpu().reshape(-1, Cout)
        want = torch.stack([yd.sum(0), (yd * yd).sum(0)], 1)
        assert relerr(part.sum(0), want) < 1e-5
        nout = min(3, nparts)
        folded = torch.full((nout, Cout, 2), 7.0, dtype=torch.float32, device=DEV)
        call("vamd_fold_partials", P(part), nparts, 2 * Cout, nout, P(folded), S())
        assert relerr(folded.sum(0), want) < 1e-5
        assert relerr(folded[0], part[0::nout].sum(0)) < 1e-6

    if up and lib.vamd_conv2d_fwd_up4_supported(C.byref(d)):
        # the same forward as four 2x2 class convolutions over the stored input
        wu = torch.full((4, Cout, 2, 2, Cin), 7.0, dtype=tdt, device=DEV)
        call("vamd_repack_weight_up4_fwd", P(w64.detach().float().to(DEV).contiguous()), P(wu), Cout, Cin, dt, S())
        y4 = torch.full((B, Ho, Wo, Cout), 7.0, dtype=tdt, device=DEV)
        call("vamd_conv2d_fwd_up4", C.byref(d), P(xn), P(wu), None, None, None, None, P(y4), None, S())
        assert relerr(y4.permute(0, 3, 1, 2), y64.detach()) < TOL_OUT[dts]
        np4 = lib.vamd_conv2d_fwd_up4_stats_parts(C.byref(d))
        part4 = torch.full((np4, Cout, 2), 7.0, dtype=torch.float32, device=DEV)
        call("vamd_conv2d_fwd_up4", C.byref(d), P(xn), P(wu), P(bias.to(DEV)), P(temb.to(DEV)), P(nhwc(res)), P(nhwc(res2)),
             P(y4), P(part4), S())
        assert relerr(y4.permute(0, 3, 1, 2), full64.detach()) < TOL_OUT[dts]
        yd4 = y4.double().cpu().reshape(-1, Cout)
        assert relerr(part4.sum(0), torch.stack([yd4.sum(0), (yd4 * yd4).sum(0)], 1)) < 1e-5

    dy, dy64 = rnd((B, Cout, Ho, Wo), tdt, gen)
    gx64, gw64 = torch.autograd.grad(y64, [x64, w64], dy64)
    add, add64 = rnd((B, Cin, H, W), tdt, gen)
    dyn = nhwc(dy)
    dx = torch.empty((B, H, W, Cin), dtype=tdt, device=DEV)
    nbytes = lib.vamd_conv2d_dgrad_workspace(C.byref(d))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=DEV)
    call("vamd_conv2d_dgrad", C.byref(d), P(dyn), P(wd), None, P(dx), P(ws), nbytes, S())
    assert relerr(dx.permute(0, 3, 1, 2), gx64) < TOL_OUT[dts]
    call("vamd_conv2d_dgrad", C.byref(d), P(dyn), P(wd), P(nhwc(add)), P(dx), P(ws), nbytes, S())
    assert relerr(dx.permute(0, 3, 1, 2), gx64 + add64) < TOL_OUT[dts]

    if up and lib.vamd_conv2d_dgrad_up4_supported(C.byref(d)):
        # the same gradient as one stride-2 convolution with the combined 4x4 filter
        w4 = torch.full((Cin, 4, 4, Cout), 7.0, dtype=tdt, device=DEV)
        call("vamd_repack_weight_up4", P(w64.detach().float().to(DEV).contiguous()), P(w4), Cout, Cin, dt, S())
        sets = [[2], [1, 2], [0, 1], [0]]
        g4 = torch.stack([torch.stack([sum(w64.detach()[:, :, r, q] for r in sets[t] for q in sets[u]) for u in range(4)], 0)
                          for t in range(4)], 0)                     # [4][4][Cout][Cin]
        assert relerr(w4.permute(1, 2, 3, 0), g4) < (1e-6 if dts == "f32" else 3e-3)
        dx4 = torch.full((B, H, W, Cin), 7.0, dtype=tdt, device=DEV)
        call("vamd_conv2d_dgrad_up4", C.byref(d), P(dyn), P(w4), None, P(dx4), S())
        assert relerr(dx4.permute(0, 3, 1, 2), gx64) < TOL_OUT[dts]
        call("vamd_conv2d_dgrad_up4", C.byref(d), P(dyn), P(w4), P(nhwc(add)), P(dx4), S())
        assert relerr(dx4.permute(0, 3, 1, 2), gx64 + add64) < TOL_OUT[dts]

    dwk = torch.full((Cout, k, k, Cin), 7.0, dtype=torch.float32, device=DEV)   # must be overwritten
    wbytes = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
    wws = torch.empty(max(wbytes, 16), dtype=torch.uint8, device=DEV)
    dbk = torch.full((Cout,), 7.0, dtype=torch.float32, device=DEV)             # bias gradient, overwritten too
    call("vamd_conv2d_wgrad", C.byref(d), P(dyn), P(xn), P(dwk), P(dbk), P(wws), wbytes, S())
    assert relerr(dbk, dy64.sum((0, 2, 3))) < TOL_F32OUT[dts]
    dwk2 = torch.full((Cout, k, k, Cin), 7.0, dtype=torch.float32, device=DEV)  # without the bias output
    call("vamd_conv2d_wgrad", C.byref(d), P(dyn), P(xn), P(dwk2), None, P(wws), wbytes, S())
    fixed_order = bool(lib.vamd_conv2d_uses_mfma(C.byref(d), 2))    # MFMA path: partial sums added in a fixed order
    same = torch.equal if fixed_order else (lambda a, b: relerr(a, b) < 1e-6)
    assert same(dwk2, dwk)
    dwo = torch.full((Cout, Cin, k, k), 7.0, dtype=torch.float32, device=DEV)   # torch layout straight from the kernel
    dbo = torch.full((Cout,), 7.0, dtype=torch.float32, device=DEV)
    call("vamd_conv2d_wgrad_oihw", C.byref(d), P(dyn), P(xn), P(dwo), P(dbo), P(wws), wbytes, S())
    assert same(dwo, dwk.permute(0, 3, 1, 2).contiguous()) and same(dbo, dbk)
    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=DEV)
    call("vamd_unpack_wgrad", P(dwk), P(dw), Cout, Cin, k, S())
    assert relerr(dw, gw64) < TOL_F32OUT[dts]
    assert relerr(dwk.permute(0, 3, 1, 2), gw64) < TOL_F32OUT[dts]


# Shapes with at least 512 patches of 256 pixels: served by the weight-gradient kernel that keeps the gradient tile in
# registers.  B, H, W, Cin, Cout
WGRAD_RESIDENT_CASES = [
    (8, 128, 128, 64, 64),        # 8 x 32 patches, whole
    (9, 120, 130, 64, 64),        # 8 x 32 patches, ragged right and bottom edges
    (16, 112, 112, 64, 64),       # 16 x 16 patches
    (11, 104, 110, 64, 64),       # 16 x 16 patches, ragged
    (8, 128, 128, 64, 128),       # two output-channel tiles
    (8, 128, 128, 128, 64),       # two input-channel tiles
    (4, 56, 56, 256, 256),        # sixteen tile pairs, 16 workgroups each
    (8, 28, 28, 512, 512),        # 64 pairs, ragged patches (28 of 32 rows and columns)
    (16, 14, 14, 512, 512),       # one 16 x 16 patch per image, 196 of 256 pixels
    (3, 30, 20, 192, 128),        # six pairs: 42 workgroups each, fewer patches than that
]


@pytest.mark.parametrize("case", WGRAD_RESIDENT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_resident(case):
    """Weight and bias gradient of 3 x 3 / stride 1 layers with 64-channel tiles at sizes the resident-gradient kernel takes,
    against fp64; with and without the bias output, in both filter layouts, twice (bit-reproducible)."""
    B, H, W, Cin, Cout = case
    tdt, dt = DT["bf16"]
    gen = torch.Generator().manual_seed(sum(case))
    x, x64 = rnd((B, Cin, H, W), tdt, gen)
    dy, dy64 = rnd((B, Cout, H, W), tdt, gen)
    w64 = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, requires_grad=True)
    (gw64,) = torch.autograd.grad(F.conv2d(x64, w64, None, padding=1), [w64], dy64)
    d = ConvDesc(B, H, W, Cin, Cout, 3, 1, 1, 0, dt, _lib.ALGO_MFMA)
    xn, dyn = nhwc(x), nhwc(dy)
    wbytes = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
    wws = torch.empty(max(wbytes, 16), dtype=torch.uint8, device=DEV)
    dwk = torch.full((Cout, 3, 3, Cin), 7.0, dtype=torch.float32, device=DEV)
    dbk = torch.full((Cout,), 7.0, dtype=torch.float32, device=DEV)
    call("vamd_conv2d_wgrad", C.byref(d), P(dyn), P(xn), P(dwk), P(dbk), P(wws), wbytes, S())
    assert relerr(dwk.permute(0, 3, 1, 2), gw64) < TOL_F32OUT["bf16"]
    assert relerr(dbk, dy64.sum((0, 2, 3))) < TOL_F32OUT["bf16"]
    wws.fill_(0x7f)                                                  # nothing of an earlier call may be read back
    dwo = torch.full((Cout, Cin, 3, 3), 7.0, dtype=torch.float32, device=DEV)
    call("vamd_conv2d_wgrad_oihw", C.byref(d), P(dyn), P(xn), P(dwo), None, P(wws), wbytes, S())
    assert torch.equal(dwo, dwk.permute(0, 3, 1, 2).contiguous())
    dbo = torch.full((Cout,), 7.0, dtype=torch.float32, device=DEV)
    call("vamd_conv2d_wgrad_oihw", C.byref(d), P(dyn), P(xn), P(dwo), P(dbo), P(wws), wbytes, S())
    assert torch.equal(dwo, dwk.permute(0, 3, 1, 2).contiguous()) and torch.equal(dbo, dbk)


# Shapes whose forward runs on the resident-filter kernel and whose weight gradient on the resident-gradient kernel (at
# least 512 patches of 256 pixels, 64 input channels).  B, H, W, Cout
PRENORM_CASES = [
    (8, 128, 128, 64),            # 8 x 32 patches, whole
    (9, 120, 130, 64),            # ragged right and bottom edges: padding must stay zero after the affine map
    (16, 112, 112, 64),           # 16 x 16 patches
    (11, 104, 110, 128),          # two column tiles, ragged 16 x 16 patches
]


@pytest.mark.parametrize("case", PRENORM_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_with_input_normalised_in_the_kernel(case):
    """vamd_conv2d_fwd_prenorm / vamd_conv2d_wgrad_prenorm_oihw (BatchNorm + LeakyReLU of the input applied to the staged
    patch in LDS) against vamd_scale_shift_act followed by the plain calls: bit-identical outputs, statistics and
    gradients; and against fp64 on the normalised tensor."""
    B, H, W, Cout = case
    Cin = 64
    tdt, dt = DT["bf16"]
    gen = torch.Generator().manual_seed(sum(case))
    x, x64 = rnd((B, H, W, Cin), tdt, gen)
    dy, dy64 = rnd((B, H, W, Cout), tdt, gen)
    res, _ = rnd((B, H, W, Cout), tdt, gen)
    w64 = torch.randn((Cout, Cin, 3, 3), generator=gen, dtype=torch.float64) / 24.0
    wf, _ = pack_weights(w64, tdt, dt)
    bias = torch.randn(Cout, generator=gen).to(DEV)
    temb = torch.randn((B, Cout), generator=gen).to(DEV)
    scale = (torch.rand(Cin, generator=gen) + 0.5).to(DEV)
    shift = torch.randn(Cin, generator=gen).to(DEV)               # non-zero: act(shift) != 0 would leak into the padding
    slope = 0.01
    d = ConvDesc(B, H, W, Cin, Cout, 3, 1, 1, 0, dt, _lib.ALGO_AUTO)
    assert lib.vamd_conv2d_prenorm_supported(C.byref(d)) == 1
    a = torch.empty_like(x)
    call("vamd_scale_shift_act", P(x), P(scale), P(shift), None, slope, 1.0, P(a), 1, B * H * W, Cin, dt, S())
    nparts = lib.vamd_conv2d_stats_parts(C.byref(d))
    assert nparts > 0
    y0, y1 = torch.empty((B, H, W, Cout), dtype=tdt, device=DEV), torch.empty((B, H, W, Cout), dtype=tdt, device=DEV)
    st0 = torch.zeros((nparts, Cout, 2), dtype=torch.float32, device=DEV)
    st1 = torch.zeros_like(st0)
    call("vamd_conv2d_fwd_stats", C.byref(d), P(a), P(wf), P(bias), P(temb), P(res), None, P(y0), P(st0), S())
    call("vamd_conv2d_fwd_prenorm", C.byref(d), P(x), P(scale), P(shift), slope, P(wf), P(bias), P(temb), P(res), None, P(y1),
         P(st1), S())
    assert torch.equal(y0, y1) and torch.equal(st0, st1)
    y2 = torch.empty_like(y0)
    call("vamd_conv2d_fwd_prenorm", C.byref(d), P(x), P(scale), P(shift), slope, P(wf), None, None, None, None, P(y2), None, S())
    a64 = a.double().cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(a64, wf.double().cpu().permute(0, 3, 1, 2), None, padding=1).permute(0, 2, 3, 1)
    assert relerr(y2, ref) < TOL_OUT["bf16"]
    # weight gradient
    wbytes = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
    wws = torch.empty(max(wbytes, 16), dtype=torch.uint8, device=DEV)
    g0 = torch.full((Cout, Cin, 3, 3), 7.0, dtype=torch.float32, device=DEV)
    g1 = torch.full_like(g0, 3.0)
    b0, b1 = torch.full((Cout,), 7.0, device=DEV), torch.full((Cout,), 3.0, device=DEV)
    call("vamd_conv2d_wgrad_oihw", C.byref(d), P(dy), P(a), P(g0), P(b0), P(wws), wbytes, S())
    wws.fill_(0x7f)
    call("vamd_conv2d_wgrad_prenorm_oihw", C.byref(d), P(dy), P(x), P(scale), P(shift), slope, P(g1), P(b1), P(wws), wbytes, S())
    assert torch.equal(g0, g1) and torch.equal(b0, b1)
    w0 = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, requires_grad=True)
    (gw64,) = torch.autograd.grad(F.conv2d(a64, w0, None, padding=1), [w0], dy64.permute(0, 3, 1, 2))
    assert relerr(g1, gw64) < TOL_F32OUT["bf16"]


def test_prenorm_refuses_what_it_cannot_do():
    dt = DT["bf16"][1]
    for shape in [(2, 16, 16, 64, 64, 3, 1), (8, 128, 128, 128, 128, 3, 1), (8, 128, 128, 64, 64, 1, 1), (8, 128, 128, 64, 64, 3, 2)]:
        B, H, W, Cin, Cout, k, stride = shape
        d = ConvDesc(B, H, W, Cin, Cout, k, stride, 1 if k == 3 else 0, 0, dt, _lib.ALGO_AUTO)
        assert lib.vamd_conv2d_prenorm_supported(C.byref(d)) == 0
        t = torch.zeros(B * H * W * max(Cin, Cout), dtype=torch.bfloat16, device=DEV)
        c = torch.zeros(max(Cin, Cout), dtype=torch.float32, device=DEV)
        rc = lib.vamd_conv2d_fwd_prenorm(C.byref(d), P(t), P(c), P(c), 0.01, P(t), None, None, None, None, P(t), None, S())
        assert rc == -2 or rc < 0, rc
    d = ConvDesc(8, 128, 128, 64, 64, 3, 1, 1, 0, DT["f32"][1], _lib.ALGO_AUTO)
    assert lib.vamd_conv2d_prenorm_supported(C.byref(d)) == 0


SHORT_CASES = [
    # B, H, W, Cin (of the 3x3), Cout, Cin2 (of the 1x1 shortcut), k
    (2, 12, 12, 64, 64, 128, 3),       # 128-row tile, 64 output channels
    (1, 20, 26, 128, 128, 64, 3),      # 256 x 128 tile
    (1, 24, 48, 128, 128, 256, 3),
    (2, 18, 18, 256, 256, 128, 3),     # 256 x 256 tile when forced
    (1, 9, 11, 512, 512, 256, 3),
    (2, 7, 9, 64, 192, 64, 3),         # partial channel tile, ragged rows
    (2, 10, 10, 64, 64, 64, 1),        # 1x1 + 1x1
]


@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("case", SHORT_CASES)
def test_conv_fwd_with_shortcut_source(case, dts):
    """conv(x; w) + conv1x1(x2; w2) in one launch against the fp64 sum and against the two-launch form."""
    B, H, W, Cin, Cout, Cin2, k = case
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    d = ConvDesc(B, H, W, Cin, Cout, k, 1, 1 if k == 3 else 0, 0, dt, _lib.ALGO_AUTO)
    if dts == "bf16" and (Cin, Cout, k) == (64, 64, 3) and os.environ.get("VAMD_C64_HALO_MIN") == "1":
        assert lib.vamd_conv2d_fwd_short_supported(C.byref(d), Cin2) == 0
        pytest.skip("the resident-filter kernel is forced for this shape; it has no second source")
    assert lib.vamd_conv2d_fwd_short_supported(C.byref(d), Cin2) == 1
    assert lib.vamd_conv2d_fwd_short_supported(C.byref(d), Cin2 + 8) == 0
    x, x64 = rnd((B, Cin, H, W), tdt, gen)
    x2, x264 = rnd((B, Cin2, H, W), tdt, gen)
    _, w64 = rnd((Cout, Cin, k, k), tdt, gen, 1.0 / math.sqrt(Cin * k * k))
    _, w264 = rnd((Cout, Cin2, 1, 1), tdt, gen, 1.0 / math.sqrt(Cin2))
    wf, _ = pack_weights(w64, tdt, dt)
    wf2, _ = pack_weights(w264, tdt, dt)
    bias = torch.randn(Cout, generator=gen)
    temb = torch.randn(B, Cout, generator=gen)
    res2, res264 = rnd((B, Cout, H, W), tdt, gen)
    want = (conv_ref(x64, w64, 1, 0, k) + F.conv2d(x264, w264) + bias.double()[None, :, None, None]
            + temb.double()[:, :, None, None] + res264)
    xn, x2n = nhwc(x).contiguous(), nhwc(x2).contiguous()
    nparts = lib.vamd_conv2d_stats_parts(C.byref(d))
    assert nparts > 0
    part = torch.full((nparts, Cout, 2), 7.0, dtype=torch.float32, device=DEV)
    y = torch.full((B, H, W, Cout), 7.0, dtype=tdt, device=DEV)
    call("vamd_conv2d_fwd_short", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), P(temb.to(DEV)), P(x2n), P(wf2), Cin2,
         P(nhwc(res2)), P(y), P(part), S())
    assert relerr(y.permute(0, 3, 1, 2), want) < TOL_OUT[dts]
    yd = y.double().cpu().reshape(-1, Cout)
    assert relerr(part.sum(0), torch.stack([yd.sum(0), (yd * yd).sum(0)], 1)) < 1e-5
    # without statistics and extras
    y0 = torch.full((B, H, W, Cout), 7.0, dtype=tdt, device=DEV)
    call("vamd_conv2d_fwd_short", C.byref(d), P(xn), P(wf), None, None, P(x2n), P(wf2), Cin2, None, P(y0), None, S())
    assert relerr(y0.permute(0, 3, 1, 2), conv_ref(x64, w64, 1, 0, k) + F.conv2d(x264, w264)) < TOL_OUT[dts]
    # the two-launch form rounds the shortcut to the compute dtype in between: equal to that rounding
    d1 = ConvDesc(B, H, W, Cin2, Cout, 1, 1, 0, 0, dt, _lib.ALGO_AUTO)
    sc = torch.empty((B, H, W, Cout), dtype=tdt, device=DEV)
    call("vamd_conv2d_fwd", C.byref(d1), P(x2n), P(wf2), None, None, None, None, P(sc), S())
    y1 = torch.empty_like(y0)
    call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), None, None, P(sc), None, P(y1), S())
    assert relerr(y0, y1) < (1e-5 if dts == "f32" else 1e-2)


def test_conv_rejects_bad_arguments():
    d = ConvDesc(1, 4, 4, 8, 8, 5, 1, 2, 0, _lib.F32, 0)
    x = torch.zeros(16, device=DEV)
    rc = lib.vamd_conv2d_fwd(C.byref(d), P(x), P(x), None, None, None, None, P(x), S())
    assert rc < 0 and b"3x3" in lib.vamd_last_error()
    d = ConvDesc(1, 4, 4, 3, 3, 3, 1, 1, 0, _lib.BF16, _lib.ALGO_MFMA)
    rc = lib.vamd_conv2d_fwd(C.byref(d), P(x), P(x), None, None, None, None, P(x), S())
    assert rc < 0 and b"MFMA" in lib.vamd_last_error()
    rc = lib.vamd_conv2d_fwd(C.byref(d), None, P(x), None, None, None, None, P(x), S())
    assert rc < 0


def padded(t64, ld, tdt):
    """Zero-pad the last axis of a CPU tensor to ld and upload in the compute dtype."""
    out = torch.zeros(t64.shape[:-1] + (ld,), dtype=tdt)
    out[..., :t64.shape[-1]] = t64.to(tdt)
    return out.to(DEV)


@pytest.mark.parametrize("algo", [_lib.ALGO_DIRECT, _lib.ALGO_MFMA])
@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 64, 64, 32), (2, 196, 196, 64), (2, 100, 40, 128), (5, 16, 200, 32),
                                   (2, 196, 64, 196), (3, 50, 72, 30)])
def test_bmm_nt_tn(shape, dts, algo):
    """Batched NT / TN products with rows padded to 16-byte multiples (the 14x14 attention level has
    196 positions: 196 is not a multiple of 8)."""
    batch, M, N, K = shape
    tdt, dt = DT[dts]
    up8 = lambda v: (v + 7) // 8 * 8
    gen = torch.Generator().manual_seed(M + N)
    _, A64 = rnd((batch, M, K), tdt, gen)
    _, B64 = rnd((batch, N, K), tdt, gen)
    lda = ldb = up8(K)
    ldc = up8(N)
    A, Bm = padded(A64, lda, tdt), padded(B64, ldb, tdt)
    ref = 0.37 * torch.bmm(A64, B64.transpose(1, 2))
    out = torch.zeros((batch, M, ldc), dtype=tdt, device=DEV)
    call("vamd_bmm_nt", P(A), P(Bm), P(out), batch, M, N, K, lda, ldb, ldc, 0.37, dt, 0, algo, S())
    assert relerr(out[..., :N], ref) < TOL_OUT[dts]
    assert float(out[..., N:].abs().max()) == 0.0 if ldc > N else True
    out32 = torch.zeros((batch, M, ldc), dtype=torch.float32, device=DEV)
    call("vamd_bmm_nt", P(A), P(Bm), P(out32), batch, M, N, K, lda, ldb, ldc, 0.37, dt, 1, algo, S())
    assert relerr(out32[..., :N], ref) < TOL_F32OUT[dts]
    # TN: C[m][n] = sum_k At[k][m] Bt[k][n]
    _, At64 = rnd((batch, K, M), tdt, gen)
    _, Bt64 = rnd((batch, K, N), tdt, gen)
    At, Bt = padded(At64, up8(M), tdt), padded(Bt64, up8(N), tdt)
    ref = torch.bmm(At64.transpose(1, 2), Bt64)
    outt = torch.zeros((batch, M, ldc), dtype=tdt, device=DEV)
    call("vamd_bmm_tn", P(At), P(Bt), P(outt), batch, M, N, K, up8(M), up8(N), ldc, 1.0, dt, algo, S())
    assert relerr(outt[..., :N], ref) < TOL_OUT[dts]


@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(4, 9, 9, 64), (3, 16, 16, 3), (2, 7, 5, 96), (2, 4, 4, 512)])
def test_batchnorm_act_fwd_bwd(shape, dts):
    B, H, W, Cc = shape
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(Cc)
    x, x64 = rnd((B, Cc, H, W), tdt, gen, 2.0)
    x = nhwc(x + 0)
    x64 = x64.requires_grad_(True)
    gamma = (torch.rand(Cc, generator=gen) + 0.5)
    beta = torch.randn(Cc, generator=gen)
    rm, rv = torch.randn(Cc, generator=gen), torch.rand(Cc, generator=gen) + 0.5
    mask, mask64 = rnd((B, Cc, H, W), tdt, gen)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    MS = 1.25                                    # fp32 factor applied on top of the mask (dropout's 1/(1-p))
    y64 = F.leaky_relu(F.batch_norm(x64, rm64, rv64, g64, b64, True, 0.1, 1e-5), 0.01) * mask64 * MS

    R = B * H * W
    nchunk = lib.vamd_stats_chunks(R, 1)
    part = torch.empty((1, nchunk, Cc, 2), dtype=torch.float32, device=DEV)
    call("vamd_chan_stats", P(x), P(part), 1, R, Cc, dt, S())
    coef = torch.empty((4, Cc), dtype=torch.float32, device=DEV)
    gm, bt, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    call("vamd_bn_finalize", P(part), nchunk, Cc, R, P(gm), P(bt), P(rmd), P(rvd), P(nbt), 0.1, 1e-5, 1,
         P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), S())
    y = torch.empty_like(x)
    maskn = nhwc(mask)
    call("vamd_scale_shift_act", P(x), P(coef[0]), P(coef[1]), P(maskn), 0.01, MS, P(y), 1, R, Cc, dt, S())
    assert relerr(y.permute(0, 3, 1, 2), y64.detach()) < TOL_OUT[dts]
    assert relerr(rmd, rm64) < 1e-5 and relerr(rvd, rv64) < 1e-5 and int(nbt) == 1

    dy, dy64 = rnd((B, Cc, H, W), tdt, gen)
    gx, gg, gb = torch.autograd.grad(y64, [x64, g64, b64], dy64)
    add, add64 = rnd((B, Cc, H, W), tdt, gen)
    dyn = nhwc(dy)
    call("vamd_norm_bwd_stats", P(dyn), P(x), P(maskn), P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 0.01, MS,
         P(part), 1, R, Cc, dt, S())
    out = torch.empty((5, Cc), dtype=torch.float32, device=DEV)
    call("vamd_bn_bwd_finalize", P(part), nchunk, Cc, R, P(coef[0]), P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(out[4]), 0, S())
    dx = torch.empty_like(x)
    call("vamd_norm_bwd_apply", P(dyn), P(x), P(maskn), P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 0.01, MS,
         P(out[2]), P(out[3]), P(out[4]), P(nhwc(add)), None, P(dx), 1, R, Cc, dt, S())
    assert relerr(out[0], gg) < 2e-4 and relerr(out[1], gb) < 2e-4
    assert relerr(dx.permute(0, 3, 1, 2), gx + add64) < TOL_OUT[dts]

    # statistics grouped per image: the same coefficients and parameter gradients, and the per-image
    # channel sums of dx (exact ones, of the fp64 gradient) without a pass over dx
    nck = lib.vamd_stats_chunks(H * W, B)
    part3 = torch.empty((B, nck, Cc, 3), dtype=torch.float32, device=DEV)
    call("vamd_bn_bwd_stats_img", P(dyn), P(x), P(maskn), P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 0.01, MS,
         P(part3), B, H * W, Cc, dt, S())
    out2 = torch.empty((5, Cc), dtype=torch.float32, device=DEV)
    dsum = torch.empty((B, Cc), dtype=torch.float32, device=DEV)
    dtot = torch.empty((Cc,), dtype=torch.float32, device=DEV)
    call("vamd_bn_bwd_finalize_img", P(part3), nck, B, Cc, H * W, P(coef[0]), P(out2[0]), P(out2[1]), P(out2[2]),
         P(out2[3]), P(out2[4]), P(dsum), P(dtot), 0, S())
    assert relerr(out2, out) < 1e-5
    ref_sum = gx.sum(dim=(2, 3))
    scale_ref = float(gx.abs().sum(dim=(2, 3)).max())      # the sums cancel: measure against the summed magnitudes
    assert float((dsum.double().cpu() - ref_sum).abs().max()) < (2e-5 if dts == "f32" else 2e-2) * scale_ref
    assert float((dtot.double().cpu() - ref_sum.sum(0)).abs().max()) < (2e-5 if dts == "f32" else 2e-2) * scale_ref * B

    # eval mode: running statistics
    call("vamd_bn_finalize", None, 0, Cc, 0, P(gm), P(bt), P(rmd), P(rvd), None, 0.1, 1e-5, 0,
         P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), S())
    call("vamd_scale_shift_act", P(x), P(coef[0]), P(coef[1]), None, 0.01, 1.0, P(y), 1, R, Cc, dt, S())
    xe = x64.detach().clone().requires_grad_(True)
    ge, be = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ye = F.leaky_relu(F.batch_norm(xe, rmd.double().cpu(), rvd.double().cpu(), ge, be, False, 0.1, 1e-5), 0.01)
    assert relerr(y.permute(0, 3, 1, 2), ye.detach()) < TOL_OUT[dts]
    # eval-mode backward: the running statistics are constants, dx = scale * gz (fixed_stats = 1), in both
    # forms of the statistics pass; dsum is then the per-image channel sum of that dx
    gxe, gge, gbe = torch.autograd.grad(ye, [xe, ge, be], dy64)
    call("vamd_norm_bwd_stats", P(dyn), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 0.01, 1.0,
         P(part), 1, R, Cc, dt, S())
    call("vamd_bn_bwd_finalize", P(part), nchunk, Cc, R, P(coef[0]), P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(out[4]), 1, S())
    call("vamd_norm_bwd_apply", P(dyn), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 0.01, 1.0,
         P(out[2]), P(out[3]), P(out[4]), None, None, P(dx), 1, R, Cc, dt, S())
    assert relerr(out[0], gge) < 2e-4 and relerr(out[1], gbe) < 2e-4
    assert relerr(dx.permute(0, 3, 1, 2), gxe) < TOL_OUT[dts]
    call("vamd_bn_bwd_stats_img", P(dyn), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 0.01, 1.0,
         P(part3), B, H * W, Cc, dt, S())
    call("vamd_bn_bwd_finalize_img", P(part3), nck, B, Cc, H * W, P(coef[0]), P(out2[0]), P(out2[1]), P(out2[2]),
         P(out2[3]), P(out2[4]), P(dsum), P(dtot), 1, S())
    assert relerr(out2, out) < 1e-5
    ref_sum = gxe.sum(dim=(2, 3))
    scale_ref = float(gxe.abs().sum(dim=(2, 3)).max())
    assert float((dsum.double().cpu() - ref_sum).abs().max()) < (2e-5 if dts == "f32" else 2e-2) * scale_ref


@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 8, 8, 32), (2, 14, 14, 512), (2, 5, 7, 64)])
def test_groupnorm_fwd_bwd(shape, dts):
    B, H, W, Cc = shape
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(Cc + 1)
    x, x64 = rnd((B, Cc, H, W), tdt, gen, 1.5)
    x = nhwc(x)
    x64.requires_grad_(True)
    gamma, beta = torch.rand(Cc, generator=gen) + 0.5, torch.randn(Cc, generator=gen)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y64 = F.group_norm(x64, 32, g64, b64, 1e-6)
    R = H * W
    nchunk = lib.vamd_stats_chunks(R, B)
    part = torch.empty((B, nchunk, Cc, 2), dtype=torch.float32, device=DEV)
    call("vamd_chan_stats", P(x), P(part), B, R, Cc, dt, S())
    coef = torch.empty((4, B, Cc), dtype=torch.float32, device=DEV)
    gm, bt = gamma.to(DEV), beta.to(DEV)
    call("vamd_gn_finalize", P(part), nchunk, B, Cc, 32, R, P(gm), P(bt), 1e-6, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), S())
    y = torch.empty_like(x)
    call("vamd_scale_shift_act", P(x), P(coef[0]), P(coef[1]), None, 1.0, 1.0, P(y), B, R, Cc, dt, S())
    assert relerr(y.permute(0, 3, 1, 2), y64.detach()) < TOL_OUT[dts]
    dy, dy64 = rnd((B, Cc, H, W), tdt, gen)
    gx, gg, gb = torch.autograd.grad(y64, [x64, g64, b64], dy64)
    dyn = nhwc(dy)
    call("vamd_norm_bwd_stats", P(dyn), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 1.0, 1.0, P(part), B, R, Cc, dt, S())
    dgb = torch.empty((2, Cc), dtype=torch.float32, device=DEV)
    abc = torch.empty((3, B, Cc), dtype=torch.float32, device=DEV)
    call("vamd_gn_bwd_finalize", P(part), nchunk, B, Cc, 32, R, P(gm), P(coef[3]), P(dgb[0]), P(dgb[1]), P(abc[0]), P(abc[1]), P(abc[2]), S())
    dx = torch.empty_like(x)
    call("vamd_norm_bwd_apply", P(dyn), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), 1.0, 1.0,
         P(abc[0]), P(abc[1]), P(abc[2]), None, None, P(dx), B, R, Cc, dt, S())
    assert relerr(dgb[0], gg) < 2e-4 and relerr(dgb[1], gb) < 2e-4
    assert relerr(dx.permute(0, 3, 1, 2), gx) < TOL_OUT[dts]


@pytest.mark.parametrize("dts", ["f32", "bf16"])
def test_softmax_transpose_add_sumpool(dts):
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(5)
    rows, cols = 37, 196
    x = (torch.randn((rows, cols), generator=gen) * 3).to(DEV)
    p = torch.empty((rows, cols), dtype=tdt, device=DEV)
    call("vamd_softmax_rows", P(x), P(p), rows, cols, cols, dt, S())
    ref = torch.softmax(x.double().cpu(), -1)
    assert relerr(p, ref) < TOL_OUT[dts]
    dp = torch.randn((rows, cols), generator=gen).to(DEV)
    ds = torch.empty_like(p)
    call("vamd_softmax_rows_bwd", P(p), P(dp), P(ds), rows, cols, cols, 0.25, dt, S())
    p64 = p.double().cpu()
    refds = 0.25 * p64 * (dp.double().cpu() - (dp.double().cpu() * p64).sum(-1, keepdim=True))
    assert relerr(ds, refds) < TOL_OUT[dts]

    t, t64 = rnd((3, 45, 70), tdt, gen)
    tt = torch.empty((3, 70, 45), dtype=tdt, device=DEV)
    call("vamd_transpose", P(t), P(tt), 3, 45, 70, 45, dt, S())
    assert torch.equal(tt.cpu().double(), t64.transpose(1, 2))

    a, a64 = rnd((1003,), tdt, gen)
    b, b64 = rnd((1003,), tdt, gen)
    yv = torch.empty_like(a)
    call("vamd_add", P(a), P(b), P(yv), 1003, dt, S())
    assert relerr(yv, a64 + b64) < TOL_OUT[dts]

    u, u64 = rnd((2, 6, 8, 16), tdt, gen)
    sp = torch.empty((2, 3, 4, 16), dtype=tdt, device=DEV)
    call("vamd_sumpool2x2", P(u), P(sp), 2, 3, 4, 16, dt, S())
    refsp = u64.reshape(2, 3, 2, 4, 2, 16).sum((2, 4))
    assert relerr(sp, refsp) < TOL_OUT[dts]


def test_time_embedding_path():
    gen = torch.Generator().manual_seed(9)
    B, dim = 5, 32
    t = torch.rand(B, generator=gen)
    w = (2.0 * math.pi * torch.exp(torch.linspace(math.log(1.0), math.log(1000.0), dim // 2)))
    emb = torch.empty((B, dim), device=DEV)
    call("vamd_timeembed", P(t.to(DEV)), P(w.to(DEV)), P(emb), B, dim, S())
    ang = w[None, :] * t[:, None]
    ref = torch.cat([torch.sin(ang), torch.cos(ang)], 1)
    assert (emb.cpu() - ref).abs().max().item() < 2e-6
    I, O = 32, 48
    x = torch.randn((B, I), generator=gen)
    W_ = torch.randn((O, I), generator=gen) / 6
    b_ = torch.randn(O, generator=gen)
    for act in (0, 1):
        x64, W64, b64 = x.double().requires_grad_(True), W_.double().requires_grad_(True), b_.double().requires_grad_(True)
        y64 = F.linear(F.leaky_relu(x64, 0.01) if act else x64, W64, b64)
        y = torch.empty((B, O), device=DEV)
        call("vamd_linear_fwd", P(x.to(DEV)), P(W_.to(DEV)), P(b_.to(DEV)), P(y), B, I, O, act, 0.01, S())
        assert relerr(y, y64.detach()) < 1e-5
        dy = torch.randn((B, O), generator=gen)
        gx, gw, gb = torch.autograd.grad(y64, [x64, W64, b64], dy.double())
        dx = torch.ones((B, I), device=DEV)
        dW, db = torch.empty((O, I), device=DEV), torch.empty(O, device=DEV)
        call("vamd_linear_bwd", P(x.to(DEV)), P(W_.to(DEV)), P(dy.to(DEV)), P(dx), P(dW), P(db), B, I, O, act, 0.01, 1, S())
        assert relerr(dx, gx + 1.0) < 1e-5 and relerr(dW, gw) < 1e-5 and relerr(db, gb) < 1e-5


@pytest.mark.parametrize("shape", [(3, 3, 10, 6), (2, 8, 5, 7), (3, 40, 7, 7), (2, 256, 7, 7)],
                         ids=lambda v: "x".join(map(str, v)))
@pytest.mark.parametrize("dts", ["f32", "bf16"])
def test_diffusion_elementwise(dts, shape):
    """Images (3 channels: one lane per pixel) and latents (many channels: one lane per element, tiled layout
    converters; 40 channels and 49 positions are no multiples of the 32 x 32 tile)."""
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(3)
    B, Cc, H, W = shape
    x0, eps = torch.rand((B, Cc, H, W), generator=gen), torch.randn((B, Cc, H, W), generator=gen)
    t = torch.rand(B, generator=gen)
    lo, hi = torch.acos(torch.tensor(0.95)), torch.acos(torch.tensor(0.02))
    ang = lo + t * (hi - lo)
    n, s = torch.sin(ang), torch.cos(ang)
    xt = torch.empty((B, H, W, Cc), dtype=tdt, device=DEV)
    rates = torch.empty((3, B), device=DEV)
    call("vamd_qsample", P(x0.to(DEV)), P(eps.to(DEV)), P(t.to(DEV)), P(xt), P(rates[0]), P(rates[1]), P(rates[2]),
         B, Cc, H, W, dt, S())
    ref = s[:, None, None, None] * x0 + n[:, None, None, None] * eps
    assert relerr(xt.permute(0, 3, 1, 2), ref) < TOL_OUT[dts]
    assert (rates[0].cpu() - n).abs().max() < 3e-7 and (rates[1].cpu() - s).abs().max() < 3e-7
    assert (rates[2].cpu() - n * n).abs().max() < 3e-7
    r2 = torch.empty((3, B), device=DEV)
    call("vamd_schedule", P(t.to(DEV)), P(r2[0]), P(r2[1]), P(r2[2]), B, S())
    assert torch.equal(r2, rates)

    pred, pred64 = rnd((B, Cc, H, W), tdt, gen)
    predn = nhwc(pred)
    part = torch.empty(lib.vamd_l1_chunks(B * H * W), device=DEV)
    loss = torch.empty((), device=DEV)
    call("vamd_l1_loss_fwd", P(predn), P(eps.to(DEV)), P(part), P(loss), B, Cc, H, W, dt, S())
    assert abs(loss.item() - (pred64 - eps.double()).abs().mean().item()) < 1e-6
    g = torch.tensor([2.0], device=DEV)
    dp = torch.empty_like(predn)
    call("vamd_l1_loss_bwd", P(predn), P(eps.to(DEV)), P(g), P(dp), B, Cc, H, W, dt, S())
    refg = 2.0 * torch.sign(pred64 - eps.double()) / pred64.numel()
    assert relerr(dp.permute(0, 3, 1, 2), refg) < TOL_OUT[dts]

    cur = torch.randn((B, Cc, H, W), generator=gen)
    curd = torch.empty((B, H, W, Cc), device=DEV)
    tmp = torch.empty((B, H, W, Cc), dtype=tdt, device=DEV)
    call("vamd_nchw_to_nhwc", P(cur.to(DEV)), P(tmp), P(curd), B, Cc, H, W, dt, S())
    assert torch.equal(curd.permute(0, 3, 1, 2).cpu(), cur)
    back = torch.empty((B, Cc, H, W), device=DEV)
    call("vamd_nhwc_to_nchw", P(tmp), P(back), B, Cc, H, W, dt, S())
    assert relerr(back, cur) < TOL_OUT[dts]
    n2, s2 = torch.rand(B, generator=gen), torch.rand(B, generator=gen) + 0.1
    pi = torch.empty((B, Cc, H, W), device=DEV)
    nxt = torch.empty((B, H, W, Cc), dtype=tdt, device=DEV)
    call("vamd_denoise_step", P(curd), P(predn), P(n.to(DEV)), P(s.to(DEV)), P(n2.to(DEV)), P(s2.to(DEV)), P(nxt), P(pi),
         B, Cc, H, W, dt, S())
    bc = lambda v: v.double()[:, None, None, None]
    x0h = (cur.double() - bc(n) * pred64) / bc(s)
    assert relerr(pi, x0h) < 1e-5
    assert relerr(curd.permute(0, 3, 1, 2), bc(s2) * x0h + bc(n2) * pred64) < 1e-5
    assert relerr(nxt.permute(0, 3, 1, 2), bc(s2) * x0h + bc(n2) * pred64) < TOL_OUT[dts]


def test_adamw_matches_torch():
    gen = torch.Generator().manual_seed(1)
    n = 1027
    p0 = torch.randn(n, generator=gen)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.1)
    p = p0.to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        g = torch.randn(n, generator=gen)
        ref.grad = g.clone()
        opt.step()
        call("vamd_adamw", P(p), P((2 * g).to(DEV)), P(m), P(v), n, 1e-2, None, 0.9, 0.999, 1e-8, 0.1, step, 0.5, S())
    assert relerr(p, ref.detach()) < 1e-6


@pytest.mark.parametrize("l2", [0, 1])
def test_adam_with_step_count_and_rate_on_the_device(l2):
    """vamd_adam_dev (learning rate and step count read from device memory, bias corrections formed in the kernel) against
    vamd_adamw / vamd_adam_l2 with the host's values, at steps 1, 2, 10 and 1000: the same update to 2e-5 of its size (fp32 powf against the host's double)."""
    gen = torch.Generator().manual_seed(9)
    n = 4099
    for step in (1, 2, 10, 1000):
        p0 = torch.randn(n, generator=gen).to(DEV)
        g = torch.randn(n, generator=gen).to(DEV)
        m0 = (torch.randn(n, generator=gen) * 0.1).to(DEV)
        v0 = (torch.rand(n, generator=gen) * 0.01).to(DEV)
        a, b = (p0.clone(), m0.clone(), v0.clone()), (p0.clone(), m0.clone(), v0.clone())
        call("vamd_adam_l2" if l2 else "vamd_adamw", P(a[0]), P(g), P(a[1]), P(a[2]), n, 3e-3, None, 0.9, 0.999, 1e-8, 1e-2, step, 0.5, S())
        lr_dev = torch.full((1,), 3e-3, device=DEV)
        step_dev = torch.full((1,), step, dtype=torch.int64, device=DEV)
        call("vamd_adam_dev", P(b[0]), P(g), P(b[1]), P(b[2]), n, P(lr_dev), P(step_dev), 0.9, 0.999, 1e-8, 1e-2, 0.5, l2, S())
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert ((a[0] - b[0]).norm() / (a[0] - p0).norm()).item() < 2e-5, step


@pytest.mark.parametrize("dts", ["f32", "bf16"])
def test_repack_weights_batched_matches_single(dts):
    """One launch over a table of convolutions == vamd_repack_weight per convolution."""
    from vision_amd._lib import RepackItem
    dt, tdt = (_lib.F32, torch.float32) if dts == "f32" else (_lib.BF16, torch.bfloat16)
    shapes = [(64, 3, 3), (3, 64, 3), (3, 3, 1), (40, 72, 3), (128, 64, 1), (33, 31, 3), (96, 96, 3)]
    gen = torch.Generator().manual_seed(5)
    items = (RepackItem * len(shapes))()
    keep, blocks = [], 0
    for it, (O, I, k) in zip(items, shapes):
        w = torch.randn((O, I, k, k), generator=gen).to(DEV)
        wf = torch.full((O, k, k, I), 7.0, dtype=tdt, device=DEV)
        wd = torch.full((I, k, k, O), 7.0, dtype=tdt, device=DEV)
        wf1, wd1 = torch.empty_like(wf), torch.empty_like(wd)
        call("vamd_repack_weight", P(w), P(wf1), P(wd1), O, I, k, dt, S())
        it.w, it.w_fwd, it.w_dgrad = w.data_ptr(), wf.data_ptr(), wd.data_ptr()
        it.first_block, it.Cout, it.Cin, it.ksize = blocks, O, I, k
        blocks += ((O + 31) // 32) * ((I + 31) // 32)
        keep.append((w, wf, wd, wf1, wd1))
    # the 3 x 3 entries also carry the combined filters of an upsampling convolution
    up = []
    for it, (O, I, k), (w, *_rest) in zip(items, shapes, keep):
        if k != 3:
            continue
        g4, f4 = torch.full((I, 4, 4, O), 7.0, dtype=tdt, device=DEV), torch.full((4, O, 2, 2, I), 7.0, dtype=tdt, device=DEV)
        g41, f41 = torch.empty_like(g4), torch.empty_like(f4)
        call("vamd_repack_weight_up4", P(w), P(g41), O, I, dt, S())
        call("vamd_repack_weight_up4_fwd", P(w), P(f41), O, I, dt, S())
        it.w_up4_dgrad, it.w_up4_fwd = g4.data_ptr(), f4.data_ptr()
        up.append((g4, f4, g41, f41))
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(DEV)
    call("vamd_repack_weights", P(table), len(shapes), blocks, dt, S())
    torch.cuda.synchronize()
    for w, wf, wd, wf1, wd1 in keep:
        assert torch.equal(wf, wf1) and torch.equal(wd, wd1)
        assert torch.equal(wf.float(), w.permute(0, 2, 3, 1).to(tdt).float())
    assert up and all(torch.equal(g4, g41) and torch.equal(f4, f41) for g4, f4, g41, f41 in up)


@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 18, 18, 256), (3, 14, 14, 512), (2, 9, 11, 64)], ids=lambda s: "x".join(map(str, s)))
def test_three_projections_as_one_convolution(shape, dts):
    """The pieces the fused q / k / v projection of an attention block is made of: the weight re-pack writing three
    filters' input-gradient forms side by side (ld_dgrad) and their biases into one vector, the forward / input gradient
    of the 3C-wide 1x1 convolution against the three separate ones, the weight gradient split over three destinations
    (vamd_conv2d_wgrad_oihw_split3) against three separate calls, and the transpose of a channel range (vamd_transpose_ld)."""
    from vision_amd._lib import RepackItem
    B, H, W, Cc = shape
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(sum(shape))
    ws = [(torch.randn((Cc, Cc, 1, 1), generator=gen) / math.sqrt(Cc)).to(DEV) for _ in range(3)]
    bs = [torch.randn(Cc, generator=gen).to(DEV) for _ in range(3)]
    wf3 = torch.full((3 * Cc, 1, 1, Cc), 7.0, dtype=tdt, device=DEV)
    wd3 = torch.full((Cc, 1, 1, 3 * Cc), 7.0, dtype=tdt, device=DEV)
    b3 = torch.full((3 * Cc,), 7.0, device=DEV)
    singles = []
    items, blocks = (RepackItem * 3)(), 0
    for j, (it, w, b) in enumerate(zip(items, ws, bs)):
        wf1 = torch.empty((Cc, 1, 1, Cc), dtype=tdt, device=DEV)
        wd1 = torch.empty((Cc, 1, 1, Cc), dtype=tdt, device=DEV)
        call("vamd_repack_weight", P(w), P(wf1), P(wd1), Cc, Cc, 1, dt, S())
        singles.append((wf1, wd1))
        it.w, it.w_fwd = w.data_ptr(), wf3.data_ptr() + j * Cc * Cc * wf3.element_size()
        it.w_dgrad, it.ld_dgrad = wd3.data_ptr() + j * Cc * wd3.element_size(), 3 * Cc
        it.first_block, it.Cout, it.Cin, it.ksize = blocks, Cc, Cc, 1
        it.bias, it.bias_out = b.data_ptr(), b3.data_ptr() + j * Cc * 4
        blocks += ((Cc + 31) // 32) ** 2
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(DEV)
    call("vamd_repack_weights", P(table), 3, blocks, dt, S())
    assert torch.equal(wf3, torch.cat([wf1 for wf1, _ in singles], 0))
    assert torch.equal(wd3, torch.cat([wd1 for _, wd1 in singles], 3))
    assert torch.equal(b3, torch.cat(bs))
    # forward: one convolution over 3C output channels = the three side by side
    x, _ = rnd((B, H, W, Cc), tdt, gen)
    d1 = ConvDesc(B, H, W, Cc, Cc, 1, 1, 0, 0, dt, _lib.ALGO_AUTO)
    d3 = ConvDesc(B, H, W, Cc, 3 * Cc, 1, 1, 0, 0, dt, _lib.ALGO_AUTO)
    y3 = torch.empty((B, H, W, 3 * Cc), dtype=tdt, device=DEV)
    call("vamd_conv2d_fwd", C.byref(d3), P(x), P(wf3), P(b3), None, None, None, P(y3), S())
    for j in range(3):
        y1 = torch.empty((B, H, W, Cc), dtype=tdt, device=DEV)
        call("vamd_conv2d_fwd", C.byref(d1), P(x), P(singles[j][0]), P(bs[j]), None, None, None, P(y1), S())
        assert relerr(y3[..., j * Cc:(j + 1) * Cc], y1) < (1e-6 if dts == "f32" else 1e-2)
    # the transpose of a channel range
    S_, Sp = H * W, (H * W + 7) // 8 * 8
    vT = torch.zeros((B, Cc, Sp), dtype=tdt, device=DEV)
    call("vamd_transpose_ld", P(y3[..., 2 * Cc:]), P(vT), B, S_, Cc, 3 * Cc, Sp, dt, S())
    assert torch.equal(vT[:, :, :S_], y3[..., 2 * Cc:].reshape(B, S_, Cc).transpose(1, 2))
    # input gradient of the wide convolution = the sum of the three
    dy3, dy64 = rnd((B, H, W, 3 * Cc), tdt, gen)
    dx3 = torch.empty((B, H, W, Cc), dtype=tdt, device=DEV)
    nb = lib.vamd_conv2d_dgrad_workspace(C.byref(d3))
    ws_ = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    call("vamd_conv2d_dgrad", C.byref(d3), P(dy3), P(wd3), None, P(dx3), P(ws_), nb, S())
    ref = sum(dy64[..., j * Cc:(j + 1) * Cc].reshape(-1, Cc) @ singles[j][0].double().cpu().reshape(Cc, Cc) for j in range(3))
    assert relerr(dx3.reshape(-1, Cc), ref) < TOL_OUT[dts]
    # weight gradient into three destinations = three separate weight gradients, bit for bit
    if not lib.vamd_conv2d_uses_mfma(C.byref(d3), 2):
        pytest.skip("shape outside the MFMA weight-gradient kernel")
    nb3 = lib.vamd_conv2d_wgrad_workspace(C.byref(d3))
    w3 = torch.empty(max(nb3, 16), dtype=torch.uint8, device=DEV)
    gw = [torch.full((Cc, Cc, 1, 1), 7.0, device=DEV) for _ in range(3)]
    gb = [torch.full((Cc,), 7.0, device=DEV) for _ in range(3)]
    call("vamd_conv2d_wgrad_oihw_split3", C.byref(d3), P(dy3), P(x), C.byref((C.c_void_p * 3)(*[P(t) for t in gw])),
         C.byref((C.c_void_p * 3)(*[P(t) for t in gb])), P(w3), nb3, S())
    x64 = x.double().cpu().reshape(-1, Cc)
    for j in range(3):
        g64 = dy64[..., j * Cc:(j + 1) * Cc].reshape(-1, Cc).T @ x64
        assert relerr(gw[j].reshape(Cc, Cc), g64) < TOL_F32OUT[dts]
        assert relerr(gb[j], dy64[..., j * Cc:(j + 1) * Cc].reshape(-1, Cc).sum(0)) < TOL_F32OUT[dts]
    gw2 = [torch.full((Cc, Cc, 1, 1), 3.0, device=DEV) for _ in range(3)]
    call("vamd_conv2d_wgrad_oihw_split3", C.byref(d3), P(dy3), P(x), C.byref((C.c_void_p * 3)(*[P(t) for t in gw2])), None,
         P(w3), nb3, S())
    assert all(torch.equal(a, b) for a, b in zip(gw, gw2))


def test_linear_fwd_multi_matches_single():
    """Several linear layers over one input in one launch = the same layers one by one."""
    from vision_amd._lib import LinearItem
    gen = torch.Generator().manual_seed(5)
    B, I, outs = 6, 96, [64, 40, 512, 8]
    x = torch.randn((B, I), generator=gen).to(DEV)
    Ws = [torch.randn((o, I), generator=gen).to(DEV) for o in outs]
    bs = [torch.randn((o,), generator=gen).to(DEV) if k != 1 else None for k, o in enumerate(outs)]
    ys = [torch.full((B, o), 7.0, device=DEV) for o in outs]
    items, first = (LinearItem * len(outs))(), 0
    for it, W, b, y, o in zip(items, Ws, bs, ys, outs):
        it.W, it.bias, it.y, it.O, it.first = W.data_ptr(), (b.data_ptr() if b is not None else None), y.data_ptr(), o, first
        first += o
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(DEV)
    call("vamd_linear_fwd_multi", P(x), P(table), len(outs), first, B, I, 1, 0.01, S())
    for W, b, y, o in zip(Ws, bs, ys, outs):
        one = torch.empty((B, o), device=DEV)
        call("vamd_linear_fwd", P(x), P(W), P(b), P(one), B, I, o, 1, 0.01, S())
        assert torch.equal(y, one)
        ref = F.leaky_relu(x.double().cpu(), 0.01) @ W.double().cpu().T + (b.double().cpu() if b is not None else 0.0)
        assert relerr(y, ref) < 1e-5


@pytest.mark.parametrize("dts", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (3, 7, 9, 64), (2, 6, 10, 3), (1, 224, 224, 64)], ids=lambda s: "x".join(map(str, s)))
def test_pool_and_unpool_2x2(shape, dts):
    """nn.AvgPool2d((2, 2)) and nn.Upsample(scale_factor=2, mode='nearest') of the resampling blocks built without a
    convolution (reference models/ddpm.py:21,33), forward and gradient, each with the fused addend, against torch
    on the CPU in fp64.  Odd extents: the pooling drops the last row / column, its gradient leaves them zero."""
    B, H, W, Cc = shape
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((B, Cc, H, W), tdt, gen)
    Ho, Wo = H // 2, W // 2
    add, add64 = rnd((B, Cc, Ho, Wo), tdt, gen)
    dy, dy64 = rnd((B, Cc, Ho, Wo), tdt, gen)
    addx, addx64 = rnd((B, Cc, H, W), tdt, gen)
    tol = 1e-6 if dts == "f32" else 4e-3
    x64 = x64.requires_grad_(True)
    y64 = F.avg_pool2d(x64, (2, 2))
    (gx,) = torch.autograd.grad(y64, [x64], dy64)
    xn, addn, dyn, addxn = nhwc(x), nhwc(add), nhwc(dy), nhwc(addx)
    y = torch.full((B, Ho, Wo, Cc), 7.0, dtype=tdt, device=DEV)
    call("vamd_pool2x2", P(xn), P(addn), P(y), B, H, W, Cc, 0.25, dt, S())
    assert relerr(y.permute(0, 3, 1, 2), y64.detach() + add64) < tol
    call("vamd_pool2x2", P(xn), None, P(y), B, H, W, Cc, 0.25, dt, S())
    assert relerr(y.permute(0, 3, 1, 2), y64.detach()) < tol
    dx = torch.full((B, H, W, Cc), 7.0, dtype=tdt, device=DEV)
    call("vamd_unpool2x2", P(dyn), P(addxn), P(dx), B, Ho, Wo, Cc, H, W, 0.25, dt, S())
    assert relerr(dx.permute(0, 3, 1, 2), gx + addx64) < tol
    call("vamd_unpool2x2", P(dyn), None, P(dx), B, Ho, Wo, Cc, H, W, 0.25, dt, S())
    assert relerr(dx.permute(0, 3, 1, 2), gx) < tol
    # upsampling: forward = unpool(1) to exactly twice the size, gradient = pool(1)
    s, s64 = rnd((B, Cc, Ho, Wo), tdt, gen)
    s64 = s64.requires_grad_(True)
    u64 = F.interpolate(s64, scale_factor=2, mode="nearest")
    du, du64 = rnd(tuple(u64.shape), tdt, gen)
    (gs,) = torch.autograd.grad(u64, [s64], du64)
    u = torch.full((B, 2 * Ho, 2 * Wo, Cc), 7.0, dtype=tdt, device=DEV)
    call("vamd_unpool2x2", P(nhwc(s)), None, P(u), B, Ho, Wo, Cc, 2 * Ho, 2 * Wo, 1.0, dt, S())
    assert torch.equal(u.permute(0, 3, 1, 2).cpu().double(), u64.detach())
    ds = torch.full((B, Ho, Wo, Cc), 7.0, dtype=tdt, device=DEV)
    call("vamd_pool2x2", P(nhwc(du)), P(addn), P(ds), B, 2 * Ho, 2 * Wo, Cc, 1.0, dt, S())
    assert relerr(ds.permute(0, 3, 1, 2), gs + add64) < tol


# Layers with fewer tiles than CUs and long reductions: with a scratch buffer in the descriptor the filter rows are
# computed by separate workgroup sets and added by a second kernel.  B, H, W, Cin, Cout, k, stride, up
SPLITK_CASES = [
    (16, 14, 14, 512, 512, 3, 1, 0),      # the flowers / sampler-batch shape: 100 tiles, 72 K-steps
    (2, 14, 14, 512, 512, 3, 1, 0),
    (4, 32, 32, 256, 256, 3, 1, 0),       # ffhq at 4 images
    (3, 9, 11, 256, 384, 3, 1, 0),        # 48 column threads in the adding kernel: 5 row lanes, 16 threads idle
    (1, 20, 26, 256, 64, 3, 1, 0),        # 64-channel column tile
    (5, 7, 7, 512, 256, 3, 1, 0),         # ragged last block of rows
]


@pytest.mark.parametrize("dts", ["bf16", "f32"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_split_reduction(case, dts):
    """Forward (bias, time embedding, residual, skip, fused BatchNorm statistics) and accumulating input gradient with
    the reduction split by filter rows, against fp64 and against the unsplit launch (fp32 summation order differs)."""
    B, H, W, Cin, Cout, k, stride, up = case
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(sum(case) + 1)
    x, x64 = rnd((B, Cin, H, W), tdt, gen)
    _, w64 = rnd((Cout, Cin, k, k), tdt, gen, 1.0 / math.sqrt(Cin * k * k))
    wf, wd = pack_weights(w64, tdt, dt)
    bias, temb = torch.randn(Cout, generator=gen), torch.randn(B, Cout, generator=gen)
    x64 = x64.requires_grad_(True)
    y64 = conv_ref(x64, w64, stride, up, k)
    res, res64 = rnd(tuple(y64.shape), tdt, gen)
    res2, res264 = rnd(tuple(y64.shape), tdt, gen)
    dy, dy64 = rnd(tuple(y64.shape), tdt, gen)
    add, add64 = rnd((B, Cin, H, W), tdt, gen)
    want = y64.detach() + bias.double()[None, :, None, None] + temb.double()[:, :, None, None] + res64 + res264
    (gx,) = torch.autograd.grad(y64, [x64], dy64)
    xn, dyn, resn, res2n, addn = nhwc(x), nhwc(dy), nhwc(res), nhwc(res2), nhwc(add)
    scratch = torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
    tol = TOL_OUT[dts]
    outs, parts = {}, {}
    old = lib.vamd_conv_splitk(-1)
    try:
        for mode in (1, 0):
            lib.vamd_conv_splitk(mode)
            d = ConvDesc(B, H, W, Cin, Cout, k, stride, 1, up, dt, _lib.ALGO_MFMA)
            d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
            nparts = lib.vamd_conv2d_stats_parts(C.byref(d))
            assert nparts > 0
            parts[mode] = nparts
            y = torch.full((B, H, W, Cout), 7.0, dtype=tdt, device=DEV)
            part = torch.full((nparts, Cout, 2), 7.0, dtype=torch.float32, device=DEV)
            call("vamd_conv2d_fwd_stats", C.byref(d), P(xn), P(wf), P(bias.to(DEV)), P(temb.to(DEV)), P(resn), P(res2n), P(y),
                 P(part), S())
            yd = y.double().cpu().reshape(-1, Cout)
            assert relerr(part.sum(0), torch.stack([yd.sum(0), (yd * yd).sum(0)], 1)) < 1e-5, mode
            assert relerr(y.permute(0, 3, 1, 2), want) < tol, mode
            y0 = torch.full((B, H, W, Cout), 7.0, dtype=tdt, device=DEV)
            call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), None, None, None, None, P(y0), S())
            assert relerr(y0.permute(0, 3, 1, 2), y64.detach()) < tol, mode
            dx = torch.full((B, H, W, Cin), 7.0, dtype=tdt, device=DEV)
            nb = lib.vamd_conv2d_dgrad_workspace(C.byref(d))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
            call("vamd_conv2d_dgrad", C.byref(d), P(dyn), P(wd), P(addn), P(dx), P(ws), nb, S())
            assert relerr(dx.permute(0, 3, 1, 2), gx + add64) < tol, mode
            outs[mode] = (y, dx)
    finally:
        lib.vamd_conv_splitk(old)
    rows = 16
    assert parts[1] == (B * H * W + rows - 1) // rows and parts[1] != parts[0], parts      # the split form did run
    assert relerr(outs[1][0], outs[0][0].double()) < (2e-3 if dts == "bf16" else 1e-6)
    assert relerr(outs[1][1], outs[0][1].double()) < (2e-3 if dts == "bf16" else 1e-6)


ACT_CASES = [
    # B, H, W, Cin, Cout, k, stride   (which kernel serves the shape at bf16)
    (2, 20, 24, 64, 64, 3, 1),          # resident-filter kernel
    (2, 20, 24, 64, 128, 3, 1),         # ... two column tiles
    (2, 16, 16, 128, 128, 3, 1),        # 256 x 128 tiles
    (1, 24, 48, 128, 256, 3, 1),
    (4, 64, 64, 256, 256, 3, 1),        # 256 x 256 tiles (>= 160 of them need a larger batch: falls to 256 x 128 here)
    (2, 18, 18, 256, 512, 1, 1),        # 1 x 1
    (1, 18, 22, 256, 256, 3, 2),        # stride 2 (DownBlock)
    (2, 12, 12, 64, 64, 3, 2),
    (3, 9, 7, 32, 64, 3, 1),            # 128 x 64 tiles
    (16, 14, 14, 512, 512, 3, 1),       # split reduction (scratch buffer in the descriptor)
    (2, 9, 9, 64, 96, 1, 1),            # ragged column tile
]


@pytest.mark.parametrize("dts", ["bf16", "f32"])
@pytest.mark.parametrize("case", ACT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_forward_with_fused_eval_norm(case, dts):
    """vamd_conv2d_fwd_act against vamd_conv2d_fwd followed by vamd_scale_shift_act: both outputs bit-identical, with
    and without the raw output, with every epilogue term."""
    B, H, W, Cin, Cout, k, stride = case
    tdt, dt = DT[dts]
    gen = torch.Generator().manual_seed(sum(case) + 3)
    pad = 1 if k == 3 else 0
    x, _ = rnd((B, Cin, H, W), tdt, gen)
    _, w64 = rnd((Cout, Cin, k, k), tdt, gen, 1.0 / math.sqrt(Cin * k * k))
    wf, _ = pack_weights(w64, tdt, dt)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    bias, temb = torch.randn(Cout, generator=gen).to(DEV), torch.randn(B, Cout, generator=gen).to(DEV)
    res = torch.randn((B, Ho, Wo, Cout), generator=gen).to(DEV).to(tdt)
    res2 = torch.randn((B, Ho, Wo, Cout), generator=gen).to(DEV).to(tdt)
    scale = (torch.rand(Cout, generator=gen) + 0.5).to(DEV) * torch.where(torch.rand(Cout, generator=gen) < 0.2, -1.0, 1.0).to(DEV)
    shift = torch.randn(Cout, generator=gen).to(DEV)
    xn = nhwc(x)
    scratch = torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
    d = ConvDesc(B, H, W, Cin, Cout, k, stride, pad, 0, dt, _lib.ALGO_MFMA)
    d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
    assert lib.vamd_conv2d_fwd_act_supported(C.byref(d)) == 1
    for terms in ((temb, res, res2), (None, None, None), (temb, None, None), (None, res, None)):
        t_, r1, r2 = terms
        q = torch.full((B, Ho, Wo, Cout), 7.0, dtype=tdt, device=DEV)
        call("vamd_conv2d_fwd", C.byref(d), P(xn), P(wf), P(bias), P(t_), P(r1), P(r2), P(q), S())
        want = torch.full_like(q, 7.0)
        call("vamd_scale_shift_act", P(q), P(scale), P(shift), None, 0.01, 1.0, P(want), 1, B * Ho * Wo, Cout, dt, S())
        for with_raw in (True, False):
            y = torch.full_like(q, 5.0)
            raw = torch.full_like(q, 5.0) if with_raw else None
            call("vamd_conv2d_fwd_act", C.byref(d), P(xn), P(wf), P(bias), P(t_), P(r1), P(r2), P(scale), P(shift), 0.01, P(y),
                 P(raw), S())
            assert torch.equal(y, want), (terms[0] is not None, terms[1] is not None, with_raw)
            if with_raw:
                assert torch.equal(raw, q)


def test_fused_eval_norm_refuses_what_it_cannot_do():
    d = ConvDesc(2, 8, 8, 64, 64, 3, 1, 1, 1, _lib.BF16, _lib.ALGO_MFMA)       # upsampling gather
    assert lib.vamd_conv2d_fwd_act_supported(C.byref(d)) == 0
    d = ConvDesc(2, 8, 8, 3, 64, 3, 1, 1, 0, _lib.BF16, _lib.ALGO_AUTO)        # thin-channel kernel
    assert lib.vamd_conv2d_fwd_act_supported(C.byref(d)) == 0
    z = torch.zeros(64, device=DEV)
    rc = lib.vamd_conv2d_fwd_act(C.byref(d), P(z), P(z), None, None, None, None, P(z), P(z), C.c_float(0.01), P(z), None, S())
    assert rc != 0 and b"vamd_conv2d_fwd_act" in lib.vamd_last_error()


def test_linear_bwd_multi_matches_per_item_calls():
    """The gradients of several linear layers over one input in two launches: dW / dbias bit-identical to the per-layer
    calls (same summation order), dx equal to their accumulated sum up to fp32 rounding (the items are summed in table
    order inside one thread instead of launch by launch)."""
    from vision_amd._lib import LinearBwdItem
    gen = torch.Generator().manual_seed(6)
    B, I = 5, 72
    Os = [16, 40, 130, 8]
    x = torch.randn((B, I), generator=gen).to(DEV)
    Ws = [torch.randn((O, I), generator=gen).to(DEV) for O in Os]
    dys = [torch.randn((B, O), generator=gen).to(DEV) for O in Os]
    want_dx = torch.zeros((B, I), device=DEV)
    want = []
    for W_, dy in zip(Ws, dys):
        dW, db = torch.empty_like(W_), torch.empty(W_.shape[0], device=DEV)
        call("vamd_linear_bwd", P(x), P(W_), P(dy), P(want_dx), P(dW), P(db), B, I, W_.shape[0], 1, 0.01, 1, S())
        want.append((dW, db))
    got = [(torch.full_like(W_, 7.0), torch.full((W_.shape[0],), 7.0, device=DEV)) for W_ in Ws]
    items = (LinearBwdItem * len(Os))()
    tiles = 0
    for it, W_, dy, (dW, db), O in zip(items, Ws, dys, got, Os):
        it.W, it.dy, it.dW, it.dbias, it.O = W_.data_ptr(), dy.data_ptr(), dW.data_ptr(), db.data_ptr(), O
        tiles += (O + 15) // 16
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(DEV)
    dx = torch.full((B, I), 7.0, device=DEV)
    call("vamd_linear_bwd_multi", P(x), P(table), len(Os), tiles, B, I, 1, 0.01, P(dx), 0, S())
    for (dW, db), (wW, wb) in zip(got, want):
        assert torch.equal(dW, wW) and torch.equal(db, wb)
    assert relerr(dx, want_dx) < 1e-6
