"""Diagnostics (not collected): where the HOST time of a training step goes (python bench-style loop under cProfile).
    python tests/diag_host_gpu.py [preset] [batch] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vision_amd                                    # noqa: E402
from vision_amd.parallel import DataParallel         # noqa: E402
from vision_amd.presets import get_preset            # noqa: E402

preset = get_preset(sys.argv[1] if len(sys.argv) > 1 else "flowers")
B = int(sys.argv[2]) if len(sys.argv) > 2 else preset["data_config"]["batch_size"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = preset["model_config"]
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = vision_amd.get_model(cfg["name"], cfg, compute_dtype="bf16").to(dev).train()
dp = DataParallel(model)
opt = model.configure_optimizers()[0][0]
P = preset["data_config"]["patch_size"]
batch = (torch.rand((B, 3, P, P), device=dev), torch.zeros(B, dtype=torch.int64, device=dev))


def step(i):
    dp.broadcast_buffers()
    loss = model.training_step(batch, i)
    loss.backward()
    opt.step()


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
    if i == 3:
        free = (time.perf_counter() - t0) / 4          # the queue still has room: the host's own pace
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"host issue {1e3 * free:.3f} ms/step unthrottled, {1e3 * host / steps:.3f} over {steps} steps; wall {1e3 * wall / steps:.3f} ms/step")
steps = 4
# host-only cost: the same loop with the device idle between steps (nothing queued ahead)
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
