"""Micro-benchmark (not a pytest file): BatchNorm + LeakyReLU in front of a 64-channel 3x3 convolution, as a separate
vamd_scale_shift_act pass followed by the plain forward / weight gradient, against the kernels that normalise their
input patch in LDS (vamd_conv2d_fwd_prenorm / vamd_conv2d_wgrad_prenorm_oihw).

    python tests/perf_prenorm_gpu.py [B,H,W,Cout ...]
"""
import ctypes as C
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
from vision_amd import _lib                      # noqa: E402
from vision_amd._lib import ConvDesc, call, lib   # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=20):
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
    st = torch.cuda.current_stream().cuda_stream
    for cs in sys.argv[1:] or ["64,224,224,64", "64,112,112,64", "64,112,112,128", "4,1024,1024,64", "4,512,512,64"]:
        B, H, W, Cout = [int(t) for t in cs.split(",")]
        Cin = 64
        d = ConvDesc(B, H, W, Cin, Cout, 3, 1, 1, 0, _lib.BF16, 0)
        if not lib.vamd_conv2d_prenorm_supported(C.byref(d)):
            print(cs, "not supported")
            continue
        x = torch.randn((B, H, W, Cin), device=DEV).bfloat16()
        a = torch.empty_like(x)
        wf = (torch.randn((Cout, 3, 3, Cin), device=DEV) * 0.05).bfloat16()
        y = torch.empty((B, H, W, Cout), device=DEV, dtype=torch.bfloat16)
        dy = torch.randn((B, H, W, Cout), device=DEV).bfloat16()
        res = torch.randn_like(y)
        dw = torch.empty((Cout, Cin, 3, 3), device=DEV, dtype=torch.float32)
        bias, dbias = torch.randn(Cout, device=DEV), torch.empty(Cout, device=DEV)
        sc, sh = torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV)
        nparts = lib.vamd_conv2d_stats_parts(C.byref(d))
        part = torch.empty((nparts, Cout, 2), device=DEV)
        nb = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
        p = lambda t: t.data_ptr()
        t_n = timeit(lambda: call("vamd_scale_shift_act", p(x), p(sc), p(sh), None, 0.01, 1.0, p(a), 1, B * H * W, Cin, _lib.BF16, st))
        t_f = timeit(lambda: call("vamd_conv2d_fwd_stats", C.byref(d), p(a), p(wf), p(bias), None, p(res), None, p(y), p(part), st))
        t_fp = timeit(lambda: call("vamd_conv2d_fwd_prenorm", C.byref(d), p(x), p(sc), p(sh), 0.01, p(wf), p(bias), None, p(res), None,
                                   p(y), p(part), st))
        t_w = timeit(lambda: call("vamd_conv2d_wgrad_oihw", C.byref(d), p(dy), p(a), p(dw), p(dbias), p(ws), nb, st))
        t_wp = timeit(lambda: call("vamd_conv2d_wgrad_prenorm_oihw", C.byref(d), p(dy), p(x), p(sc), p(sh), 0.01, p(dw), p(dbias),
                                   p(ws), nb, st))
        print("%-18s normalise %7.1f us | fwd %7.1f -> in-kernel %7.1f us | wgrad %7.1f -> in-kernel %7.1f us | saved %7.1f us" % (
            cs, t_n, t_f, t_fp, t_w, t_wp, t_n + t_f + t_w - t_fp - t_wp), flush=True)


if __name__ == "__main__":
    main()
