"""Where a forced split-tile launch of the ping-pong kernel goes wrong (not a pytest file)."""
import ctypes as C, math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from vision_amd import _lib
from vision_amd._lib import ConvDesc, call, lib
from test_kernels_gpu import rnd, nhwc, pack_weights, relerr, DT, DEV, P, S
_lib.set_default_scratch(torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=DEV))
B, H, W, Cin, Cout, k, stride, up = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,9,11,256,256,3,1,1").split(",")]
tdt, dt = DT["bf16"]
gen = torch.Generator().manual_seed(1)
x, x64 = rnd((B, Cin, H, W), tdt, gen)
_, w64 = rnd((Cout, Cin, k, k), tdt, gen, 1.0 / math.sqrt(Cin * k * k))
wf, wd = pack_weights(w64, tdt, dt)
x64 = x64.requires_grad_(True)
xin = F.interpolate(x64, scale_factor=2, mode="nearest") if up else x64
y64 = F.conv2d(xin, w64, None, stride=stride, padding=1)
dy, dy64 = rnd(tuple(y64.shape), tdt, gen)
(gx,) = torch.autograd.grad(y64, [x64], dy64)
d = ConvDesc(B, H, W, Cin, Cout, k, stride, 1, up, dt, _lib.ALGO_MFMA)
dx = torch.empty((B, H, W, Cin), dtype=tdt, device=DEV)
nb = lib.vamd_conv2d_dgrad_workspace(C.byref(d))
ws = torch.zeros(max(nb, 16), dtype=torch.uint8, device=DEV)
for rep in range(3):
    call("vamd_conv2d_dgrad", C.byref(d), P(nhwc(dy)), P(wd), None, P(dx), P(ws), nb, S())
    torch.cuda.synchronize()
    got = dx.permute(0, 3, 1, 2).double().cpu()
    print("rep", rep, "relerr", relerr(got, gx))
    if up:
        # the un-pooled gradient on the upsampled grid, as the kernel left it in the workspace
        full = ws[:B * 2 * H * 2 * W * Cin * 2].view(torch.bfloat16).view(B * 2 * H * 2 * W, Cin).double().cpu()
        xin2 = xin.detach().requires_grad_(True)
        (gfull,) = torch.autograd.grad(F.conv2d(xin2, w64, None, padding=1), [xin2], dy64)
        ref = gfull.permute(0, 2, 3, 1).reshape(-1, Cin)
        err = (full - ref).abs()
        rows = err.max(dim=1).values
        bad = (rows > 0.05).nonzero().flatten().tolist()
        print(" bad rows", len(bad), bad[:40])
        cols = err.max(dim=0).values
        badc = (cols > 0.05).nonzero().flatten().tolist()
        print(" bad cols", len(badc), badc[:40])
