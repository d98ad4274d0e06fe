"""Diagnostics (not collected): element-wise Adam trajectory of one parameter against the golden vectors."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, sub
import vision_amd
from oracle import restatement as R
name, key = sys.argv[1], sys.argv[2]
DEV = torch.device("cuda:0")
g, cfg, B = load_golden(name)
m = vision_amd.get_model("ddpm", cfg, compute_dtype=torch.float32, kernel_algo=1)
m.load_state_dict(sub(g, "state0")); m = m.to(DEV).train()
opt = m.configure_optimizers()[0][0]
sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sub(g, "state0").items()}
names = [k for k in sd if R.is_param(k)]
mm = {k: torch.zeros_like(sd[k]) for k in names}; vv = {k: torch.zeros_like(sd[k]) for k in names}
x0 = g["x0"].to(DEV)
torch.set_printoptions(precision=3, linewidth=200, sci_mode=True)
for step in range(3):
    torch.manual_seed(100 + step); eps = torch.randn_like(g["x0"]); t = torch.rand((B, 1, 1, 1))
    loss = m.p_loss(x0, eps.to(DEV), t.to(DEV)); opt.zero_grad(); loss.backward()
    l64, g64 = R.loss_and_grads(sd, cfg, g["x0"].double(), eps.double(), t.double(), None)
    mine = dict(m.named_parameters())[key].grad.double().cpu()
    print(f"step {step} loss {loss.item():.8f} fp64 {float(l64):.8f}")
    print(" fp64 grad", g64[key].flatten()[:16]); print(" mine-fp64", (mine - g64[key]).flatten()[:16])
    psd = m.state_dict()
    print(" param diff before step (max over all keys):", max(((psd[k].double().cpu() - sd[k]).abs().max().item(), k) for k in names))
    opt.step()
    R.adamw_step({k: sd[k] for k in names}, g64, mm, vv, step + 1, cfg["learning_rate"], cfg["weight_decay"])
p0 = sub(g, "state0")[key].double()
print("update mine", (m.state_dict()[key].double().cpu() - p0).flatten()[:16]); print("update fp64", (sd[key] - p0).flatten()[:16])
print("update ref ", (sub(g, "state3")[key].double() - p0).flatten()[:16])
