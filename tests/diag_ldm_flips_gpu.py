"""Diagnostic: how many signs of (prediction - noise) differ between the device-side p_loss and the recorded reference step
on the latent-diffusion fixtures, and the gradient error over all parameters that goes with them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vision_amd                         # noqa: E402
from conftest import load_golden, sub     # noqa: E402
from oracle import ldm as O               # noqa: E402

DEV = torch.device("cuda:0")
for name in ("l0", "l1"):
    g, cfg, B = load_golden(name)
    m = vision_amd.get_model("ldm", cfg, compute_dtype=torch.float32)
    m.load_state_dict(sub(g, "state0"))
    m = m.to(DEV).train()
    loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))
    loss.backward()
    pred = m._last_pred_nhwc.float().permute(0, 3, 1, 2).cpu()
    sd = {k: v.clone() for k, v in sub(g, "state0").items()}
    ref = O.unet_forward(sd, cfg, g["x_t"], g["sched_n"] ** 2, True)
    flips = ((pred - g["eps"]).sign() != (ref - g["eps"]).sign()).sum().item()
    grads = {k: p.grad.cpu() for k, p in m.named_parameters()}
    want = sub(g, "grad")
    keys = sorted(k for k in want if not k.endswith((".block1.0.2.bias", ".linear_temb.linear.bias")))
    a = torch.cat([grads[k].double().flatten() for k in keys])
    b = torch.cat([want[k].double().flatten() for k in keys])
    print(f"{name}: {flips} sign flips of {pred.numel()}, max |pred - ref| {float((pred - ref).abs().max()):.2e}, "
          f"gradient error {float((a - b).norm() / b.norm()):.2e}", flush=True)
    # the same step evaluated by the oracle on the DEVICE's schedule values: what is left is the network's own error
    import torch.nn.functional as F
    n_dev, s_dev = m.diffusion_schedule(g["t"].to(DEV))
    n_dev, s_dev = n_dev.cpu(), s_dev.cpu()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sub(g, "state0").items() if O.is_param(k)}
    work = {k: v.clone() for k, v in sub(g, "state0").items()}
    work.update(leaves)
    pred2 = O.unet_forward(work, cfg, s_dev * g["x0"] + n_dev * g["eps"], n_dev ** 2, True)
    gs = dict(zip(leaves, torch.autograd.grad(F.l1_loss(pred2, g["eps"]), list(leaves.values()))))
    c = torch.cat([gs[k].double().flatten() for k in keys])
    print(f"    schedule: max |n - ref| {float((n_dev - g['sched_n']).abs().max()):.2e}; against the oracle on the device's "
          f"schedule values: max |pred - ref| {float((pred - pred2.detach()).abs().max()):.2e}, gradient error "
          f"{float((a - c).norm() / c.norm()):.2e}", flush=True)
