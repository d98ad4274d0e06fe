"""Micro-benchmark of single convolution entry points at full-size layer shapes (not a pytest file).

    python tests/perf_conv_gpu.py [case ...]      case = B,H,W,Cin,Cout,k,stride,up[,algo]

Prints microseconds per call and the achieved TFLOP/s / GB/s for forward, dgrad and wgrad.
PERF_EPI is a bit mask of epilogue work for the forward: 1 time embedding, 2 shortcut, 4 skip, 8 BatchNorm
statistics; with any bit set the input gradient accumulates into an existing gradient.
"""
import os
import ctypes as C
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from vision_amd import _lib                      # noqa: E402
from vision_amd._lib import ConvDesc, call, lib   # noqa: E402

DEV = torch.device("cuda:0")
DEFAULT = ["64,224,224,3,64,3,1,0", "64,224,224,64,3,3,1,0", "64,224,224,3,64,1,1,0", "64,224,224,64,3,1,1,0",
           "64,224,224,64,64,3,1,0", "64,112,112,64,64,3,1,1", "64,224,224,64,64,3,2,0", "64,112,112,128,128,3,1,0",
           "64,56,56,256,256,3,1,0", "64,28,28,512,512,3,1,0", "64,14,14,512,512,3,1,0", "64,14,14,512,512,1,1,0",
           "64,112,112,64,128,1,1,0", "64,112,112,64,128,3,1,0", "64,112,112,128,64,3,1,0"]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    if os.environ.get("PERF_SCRATCH", "1") != "0":        # let the wide kernels split tiles, as the engine does
        _lib.set_default_scratch(torch.zeros(_lib.CONV_SCRATCH_BYTES, dtype=torch.uint8, device=DEV))
    cases = sys.argv[1:] or DEFAULT
    st = torch.cuda.current_stream().cuda_stream
    for cs in cases:
        v = [int(t) for t in cs.split(",")]
        B, H, W, Cin, Cout, k, stride, up = v[:8]
        algo = v[8] if len(v) > 8 else 0
        pad = 1 if k == 3 else 0
        He, We = (2 * H, 2 * W) if up else (H, W)
        Ho, Wo = (He + 2 * pad - k) // stride + 1, (We + 2 * pad - k) // stride + 1
        d = ConvDesc(B, H, W, Cin, Cout, k, stride, pad, up, _lib.BF16, algo)
        x = torch.randn((B, H, W, Cin), device=DEV).bfloat16()
        wf = (torch.randn((Cout, k, k, Cin), device=DEV) * 0.05).bfloat16()
        wd = (torch.randn((Cin, k, k, Cout), device=DEV) * 0.05).bfloat16()
        y = torch.empty((B, Ho, Wo, Cout), device=DEV, dtype=torch.bfloat16)
        dy = torch.randn((B, Ho, Wo, Cout), device=DEV).bfloat16()
        dx = torch.empty_like(x)
        dw = torch.empty((Cout, k, k, Cin), device=DEV, dtype=torch.float32)
        bias = torch.randn(Cout, device=DEV)
        dbias = torch.empty(Cout, device=DEV)
        nb1 = lib.vamd_conv2d_dgrad_workspace(C.byref(d))
        nb2 = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
        ws = torch.empty(max(nb1, nb2, 16), dtype=torch.uint8, device=DEV)
        flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        io = 2.0 * (x.numel() + y.numel())
        epi = int(os.environ.get("PERF_EPI", "0"))
        if epi:
            nparts = max(lib.vamd_conv2d_stats_parts(C.byref(d)), 1)
            part = torch.empty((nparts, Cout, 2), device=DEV)
            temb = torch.randn((B, Cout), device=DEV)
            res, res2, add = torch.randn_like(y), torch.randn_like(y), torch.randn_like(dx)
            sel = lambda bit, t: t.data_ptr() if epi & bit else None
            if epi & 8:
                t_f = timeit(lambda: call("vamd_conv2d_fwd_stats", C.byref(d), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), sel(1, temb),
                                          sel(2, res), sel(4, res2), y.data_ptr(), part.data_ptr(), st))
            else:
                t_f = timeit(lambda: call("vamd_conv2d_fwd", C.byref(d), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), sel(1, temb),
                                          sel(2, res), sel(4, res2), y.data_ptr(), st))
            t_d = timeit(lambda: call("vamd_conv2d_dgrad", C.byref(d), dy.data_ptr(), wd.data_ptr(), add.data_ptr(), dx.data_ptr(), ws.data_ptr(), nb1, st))
        else:
            t_f = timeit(lambda: call("vamd_conv2d_fwd", C.byref(d), x.data_ptr(), wf.data_ptr(), bias.data_ptr(), None, None, None, y.data_ptr(), st))
            t_d = timeit(lambda: call("vamd_conv2d_dgrad", C.byref(d), dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), ws.data_ptr(), nb1, st))
        t_w = timeit(lambda: call("vamd_conv2d_wgrad", C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), dbias.data_ptr(), ws.data_ptr(), nb2, st))
        row = "  ".join("%s %8.1f us %7.1f TF %6.0f GB/s" % (n, t, flops / t / 1e6, io / t / 1e3)
                        for n, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)))
        print("%-34s %s" % (cs, row), flush=True)


if __name__ == "__main__":
    main()
