"""Streaming passes of the normalisation layers on the shapes of the butterfly step, against torch's copy and sum of the
same tensors: python tests/perf_norm_gpu.py [B,H,W,C ...].  Prints microseconds and TB/s of the bytes a pass must move."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vision_amd import _lib
from vision_amd._lib import lib, call

DEV = torch.device("cuda:0")
DEFAULT = ["64,224,224,64", "64,112,112,128", "64,56,56,256", "64,28,28,512", "16,224,224,64", "4,1024,1024,64"]


def timeit(fn, iters=20):
    for _ in range(3):
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
    P = lambda t: None if t is None else t.data_ptr()
    for cs in sys.argv[1:] or DEFAULT:
        B, H, W, Cc = [int(v) for v in cs.split(",")]
        R = B * H * W
        x = torch.randn((B, H, W, Cc), device=DEV).bfloat16()
        dy = torch.randn_like(x)
        y = torch.empty_like(x)
        add = torch.randn_like(x)
        coef = torch.rand((4, Cc), device=DEV) + 0.5
        abc = torch.rand((3, Cc), device=DEV)
        nb = x.numel() * 2
        n0 = lib.vamd_stats_chunks(R, 1)
        part = torch.empty((n0, Cc, 2), device=DEV)
        n2 = lib.vamd_stats_chunks(H * W, B)
        part3 = torch.empty((B, n2, Cc, 3), device=DEV)
        rows = [
            ("torch copy (1R+1W)", 2, lambda: y.copy_(x)),
            ("torch sum (1R)", 1, lambda: x.sum()),
            ("chan_stats fwd (1R)", 1, lambda: call("vamd_chan_stats", P(x), P(part), 1, R, Cc, _lib.BF16, st)),
            ("bwd stats (2R)", 2, lambda: call("vamd_norm_bwd_stats", P(dy), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]),
                                               0.2, 1.0, P(part), 1, R, Cc, _lib.BF16, st)),
            ("bwd stats per image (2R)", 2, lambda: call("vamd_bn_bwd_stats_img", P(dy), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]),
                                                         P(coef[3]), 0.2, 1.0, P(part3), B, H * W, Cc, _lib.BF16, st)),
            ("scale_shift_act (1R+1W)", 2, lambda: call("vamd_scale_shift_act", P(x), P(coef[0]), P(coef[1]), None, 0.2, 1.0, P(y), 1, R, Cc,
                                                        _lib.BF16, st)),
            ("bwd apply (2R+1W)", 3, lambda: call("vamd_norm_bwd_apply", P(dy), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]),
                                                  0.2, 1.0, P(abc[0]), P(abc[1]), P(abc[2]), None, None, P(y), 1, R, Cc, _lib.BF16, st)),
            ("bwd apply + addend (3R+1W)", 4, lambda: call("vamd_norm_bwd_apply", P(dy), P(x), None, P(coef[0]), P(coef[1]), P(coef[2]),
                                                           P(coef[3]), 0.2, 1.0, P(abc[0]), P(abc[1]), P(abc[2]), P(add), None, P(y), 1, R, Cc,
                                                           _lib.BF16, st)),
        ]
        print(f"== {cs}: {nb / 1e6:.0f} MB per tensor")
        for name, k, fn in rows:
            t = timeit(fn)
            print(f"  {name:30s} {t:8.1f} us  {k * nb / t / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
