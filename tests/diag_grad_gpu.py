"""Diagnostic (not collected): per-tensor gradient error of the HIP engine vs the exact fp64 gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from conftest import load_golden, sub
import test_model_gpu as T

name, mode = sys.argv[1], sys.argv[2]
g, cfg, B = load_golden(name)
pred64, g64, e32, ebf, ebf_all, e32_all = T.exact_gradients(name, g, cfg)
dtype = T.MODES[mode][0]
m = T.make_model(cfg, sub(g, "state0"), mode).train()
m.set_dropout_masks(T.masks_for(g, cfg, dtype))
out = m(g["x_t"].to(T.DEV), (g["sched_n"] ** 2).to(T.DEV))
print("fwd rel", T.rel(out, pred64))
loss = torch.nn.functional.l1_loss(out, g["eps"].to(T.DEV))
loss.backward()
grads = {k: p.grad for k, p in m.named_parameters()}
rows = []
tot = sum(float(g64[k].double().norm() ** 2) for k in e32) ** 0.5
for k in e32:
    d = (grads[k].detach().double().cpu() - g64[k].double()).norm().item()
    rows.append((d / tot, T.rel(grads[k], g64[k]), e32[k], k))
rows.sort(reverse=True)
for r in rows[:12]:
    print("contrib %.2e  rel %.2e  ref32 %.2e  %s" % r)
print("global", sum(r[0] ** 2 for r in rows) ** 0.5)
