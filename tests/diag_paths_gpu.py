"""Diagnostics (not collected): gradients through p_loss against gradients through forward() + F.l1_loss."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, sub, noise_keys
import vision_amd
name = sys.argv[1]
algo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DEV = torch.device("cuda:0")
g, cfg, B = load_golden(name)
res = {}
for path in ("forward", "p_loss", "forward", "p_loss"):
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=torch.float32, kernel_algo=algo)
    m.load_state_dict(sub(g, "state0")); m = m.to(DEV).train()
    if path == "forward":
        out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))
        loss = torch.nn.functional.l1_loss(out, g["eps"].to(DEV))
    else:
        loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))
    loss.backward()
    grads = {k: p.grad.double().cpu().clone() for k, p in m.named_parameters() if k not in noise_keys(g)}
    if path in res:
        print(path, "repeat max rel diff", max(((grads[k] - res[path][k]).norm() / (res[path][k].norm() + 1e-30)).item() for k in grads))
    res[path] = grads
    print(path, float(loss))
rows = sorted((((res["p_loss"][k] - res["forward"][k]).norm() / (res["forward"][k].norm() + 1e-30)).item(), k) for k in res["forward"])
for e, k in rows[::-1][:12]:
    print("%.2e %s" % (e, k))
ref = sub(g, "grad")
for path in res:
    rows = sorted((((res[path][k] - ref[k].double()).norm() / (ref[k].double().norm() + 1e-30)).item(), k) for k in res[path])
    print(path, "vs golden grads:", [(k, "%.1e" % e) for e, k in rows[::-1][:5]])
# forward values of the two paths
m = vision_amd.get_model("ddpm", cfg, compute_dtype=torch.float32, kernel_algo=algo)
m.load_state_dict(sub(g, "state0")); m = m.to(DEV).train()
eng = m._ensure_engine(); eng.debug_taps = {}
out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV)).detach()
taps_f = {k: (h.clone(), None if r is None else r.clone()) for k, (h, r) in eng.debug_taps.items()}
m.load_state_dict(sub(g, "state0")); eng.debug_taps = {}
loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))
print("pred diff", (m._last_pred_nhwc.permute(0, 3, 1, 2) - out).abs().max().item(), "of", out.abs().max().item())
for k, (h, r) in eng.debug_taps.items():
    print(k, "out diff %.2e" % (h - taps_f[k][0]).abs().max().item(), "" if r is None else "skip diff %.2e" % (r - taps_f[k][1]).abs().max().item())
