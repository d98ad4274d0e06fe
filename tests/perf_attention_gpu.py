"""Micro-benchmark of the attention core: one fused kernel against the separate score / softmax / product kernels
(not a pytest file).   python tests/perf_attention_gpu.py [B,S,C ...]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0]); sys.path.insert(0, __file__.rsplit("/", 1)[0])
from test_attention_gpu import run_fused, run_separate, DEV


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for cs in sys.argv[1:] or ["4,1024,256", "16,196,512", "64,196,512", "1,1024,256"]:
    B, S, Cc = (int(v) for v in cs.split(","))
    q, k, v = (torch.randn((B, S, Cc), device=DEV).bfloat16() for _ in range(3))
    print(f"B={B} S={S} C={Cc}: fused {timeit(lambda: run_fused(q, k, v)):7.1f} us   separate {timeit(lambda: run_separate(q, k, v)):7.1f} us "
          f"(both include the value transpose and the buffer fills of the test helpers)", flush=True)
