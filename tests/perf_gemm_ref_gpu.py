"""Practical GEMM ceiling on this card for the layer shapes of the UNet (not a pytest file): torch.matmul (hipBLASLt /
rocBLAS, bf16) on the implicit-GEMM shapes M = B*Ho*Wo, N = Cout, K = 9*Cin -- a dense GEMM without any gather, i.e. an
upper bound for what a convolution kernel of the same shape can reach."""
import torch
DEV = torch.device("cuda:0")
for name, M, N, K in (("64@224^2", 64 * 224 * 224, 64, 576), ("128@112^2", 64 * 112 * 112, 128, 1152), ("256@56^2", 64 * 56 * 56, 256, 2304),
                      ("512@28^2", 64 * 28 * 28, 512, 4608), ("512@14^2", 64 * 14 * 14, 512, 4608), ("8192^3", 8192, 8192, 8192)):
    a = torch.randn((M, K), device=DEV).bfloat16()
    b = torch.randn((N, K), device=DEV).bfloat16()
    for nm, fn in (("A @ B^T", lambda: a @ b.t()),):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name:10s} M={M:7d} N={N:4d} K={K:5d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s", flush=True)
