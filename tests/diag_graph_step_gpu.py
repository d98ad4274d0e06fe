"""Eager launches against HIP-graph replay of ONE WHOLE TRAINING STEP (objective, forward, backward, optimiser, weight
re-pack); not a pytest file.  python tests/diag_graph_step_gpu.py [preset [batch]]"""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import vision_amd
from vision_amd.presets import get_preset

DEV = torch.device("cuda:0")
preset_name = sys.argv[1] if len(sys.argv) > 1 else "flowers"
preset = get_preset(preset_name)
cfg = preset["model_config"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else preset["data_config"]["batch_size"]
P = preset["data_config"]["patch_size"]
torch.manual_seed(0)
model = vision_amd.get_model(cfg["name"], cfg, compute_dtype="bf16").to(DEV).train()
if cfg["name"] == "ldm":
    fs_cfg = get_preset("vqvaeunet")["model_config"]
    model.set_first_stage(vision_amd.get_model("vqvaeunet", fs_cfg, compute_dtype="bf16").to(DEV).eval())
opt = model.configure_optimizers()[0][0]
x = torch.rand((B, 3, P, P), device=DEV)
y = torch.zeros(B, dtype=torch.int64, device=DEV)


def step():
    loss = model.training_step((x, y), 0)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3


host, eager = timeit(step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
ghost, graph = timeit(g.replay)
print("%s B=%d training step: eager %.3f ms (host issue %.3f ms)  graph replay %.3f ms (host %.3f ms)  loss %.4f" % (
    preset_name, B, eager, host, graph, ghost, float(loss)))
