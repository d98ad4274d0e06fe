"""Eager launches against HIP-graph replay of one sampler step (eval forward + denoise update); not a pytest file."""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd
from vision_amd.presets import get_preset

DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = get_preset("butterfly")["model_config"]
model = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to(DEV).eval()
eng = model._ensure_engine()
P = cfg["patch_size"]
x = torch.randn((B, P, P, 3), device=DEV).to(eng.dtype)
t = torch.rand((B,), device=DEV)


def step():
    return eng.forward(x, t, False, None)


with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20
print("B=%d eval forward: eager %.3f ms  graph replay %.3f ms" % (B, eager * 1e3, graph * 1e3))
