"""Which torch operators (not library launches) a training step still issues: torch.profiler over three eager steps of the
butterfly preset, device kernels grouped by the operator that launched them.  python tests/diag_torch_ops_gpu.py [preset]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import vision_amd
from vision_amd.presets import get_preset
from torch.profiler import profile, ProfilerActivity

name = sys.argv[1] if len(sys.argv) > 1 else "butterfly"
preset = get_preset(name)
cfg = preset["model_config"]
B, P = preset["data_config"]["batch_size"], preset["data_config"]["patch_size"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = vision_amd.get_model(cfg["name"], cfg, compute_dtype="bf16").to(dev).train()
opt = model.configure_optimizers()[0][0]
batch = (torch.rand((B, 3, P, P)).to(dev), torch.zeros(B, dtype=torch.int64, device=dev))


def step(i):
    loss = model.training_step(batch, i)
    loss.backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows[:25]:
    print(f"{e.key:40s} calls/step {e.count / 3:6.1f}  device us/step {e.device_time_total / 3:8.1f}")
print("-- memory copies and fills (any origin)")
for e in sorted(prof.key_averages(), key=lambda e: -e.count):
    if "emcpy" in e.key or "emset" in e.key or "copyBuffer" in e.key or "fillBuffer" in e.key:
        print(f"{e.key[:70]:70s} calls/step {e.count / 3:6.1f}  device us/step {e.device_time_total / 3:8.1f}")
print("-- host-side ops that contain them")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    step(7)
    torch.cuda.synchronize()
seen = {}
for ev in prof2.events():
    if ev.name in ("hipMemcpyAsync", "hipMemcpyWithStream", "hipMemcpyDtoDAsync", "hipMemsetAsync") or "emcpy" in ev.name:
        st = [s for s in (ev.stack or []) if "vision_amd" in s or "bench" in s or "tests/" in s]
        key = (ev.name, st[0] if st else "?")
        seen[key] = seen.get(key, 0) + 1
for (n, where), c in sorted(seen.items(), key=lambda kv: -kv[1])[:20]:
    print(f"{c:4d} x {n:24s} {where}")
