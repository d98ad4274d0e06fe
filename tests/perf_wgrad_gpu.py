"""Timing (not collected): weight-gradient launches of the 64-channel layers, resident-gradient kernel vs the generic one
(VAMD_WGRAD_HALO=0 in a second process).  python tests/perf_wgrad_gpu.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd import _lib                      # noqa: E402
from vision_amd._lib import ConvDesc, call, lib   # noqa: E402

DEV = torch.device("cuda:0")
CASES = [(64, 224, 224, 64, 64), (64, 112, 112, 64, 64), (64, 112, 112, 64, 128), (64, 112, 112, 128, 64),
         (64, 112, 112, 128, 128), (64, 56, 56, 128, 128), (64, 56, 56, 128, 256), (64, 56, 56, 256, 128), (64, 56, 56, 256, 256),
         (64, 28, 28, 256, 256), (64, 28, 28, 256, 512), (64, 28, 28, 512, 256), (64, 28, 28, 512, 512), (64, 14, 14, 512, 512),
         (16, 224, 224, 64, 64), (16, 112, 112, 128, 128), (16, 56, 56, 256, 256), (16, 28, 28, 512, 512), (16, 14, 14, 512, 512),
         (4, 1024, 1024, 64, 64), (4, 512, 512, 64, 64), (4, 256, 256, 64, 128), (4, 256, 256, 128, 128), (4, 128, 128, 128, 128),
         (4, 64, 64, 256, 256), (4, 32, 32, 256, 256), (4, 16, 16, 256, 256)]
S = torch.cuda.current_stream().cuda_stream
for B, H, W, Cin, Cout in CASES:
    x = torch.randn((B, H, W, Cin), device=DEV).bfloat16()
    dy = torch.randn((B, H, W, Cout), device=DEV).bfloat16()
    d = ConvDesc(B, H, W, Cin, Cout, 3, 1, 1, 0, _lib.BF16, _lib.ALGO_MFMA)
    nb = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=DEV)
    db = torch.empty(Cout, dtype=torch.float32, device=DEV)
    run = lambda: call("vamd_conv2d_wgrad_oihw", C.byref(d), x.data_ptr(), dy.data_ptr() if False else dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, S)
    # argument order of the entry point: (desc, dy, x, dw, dbias, workspace, bytes, stream)
    run = lambda: call("vamd_conv2d_wgrad_oihw", C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, S)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * Cin * Cout * 9
    print(f"wgrad {B}x{H}x{W} {Cin}->{Cout}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s (kernel + reduce)  HBM-in {(B*H*W*(Cin+Cout)*2)/ms/1e9:6.2f} TB/s", flush=True)
