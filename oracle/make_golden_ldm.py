"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/{l0,l1}.safetensors.

Runs the reference `LatentDiffusion` (models/ldm.py, loaded by oracle/ref_loader.py) on two tiny configurations and
records what oracle/make_golden.py records for the DDPM model: state, latents, the two draws of p_loss, schedule,
noisy latents, eval forward, training-mode loss and every parameter gradient, BatchNorm statistics, three AdamW steps,
validation loss and fixed-noise sampling.  Only runs where the reference is mounted (the build container):

    python -m oracle.make_golden_ldm

The reference class cannot be constructed as shipped: its constructor reads `self.attn_resolutions`, which nothing
sets (models/ldm.py:179, :206).  The file is left as it is; this script supplies that one attribute on the class from
the configuration's `attn_res` before constructing it, and sets `curr_device` (read by the schedule, models/ldm.py:250,
assigned only inside the step methods) on the instance.  Everything recorded is computed by the reference's own code.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

from oracle.make_golden import _draw
from oracle.ref_loader import load_reference_ldm

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

BASE = dict(name="ldm", learning_rate=1e-3, manual_seed=42, with_conv=True, dropout=0.0, sampling_period=1,
            weight_decay=1e-2, diffusion_steps=4, scheduler_gamma=0.99, grid=2)
CONFIGS = {
    # attention in the down path, the middle and the up path; in != out blocks both ways; one resampling level
    "l0": dict(BASE, h_dims=[32, 32, 64], down_index=[0, 1, 0], attn_res=[8], in_res=8),
    # the shape of latents of the shipped first stage: 7 x 7 maps (49 positions: not a multiple of 8), no resampling
    "l1": dict(BASE, h_dims=[32, 32, 32], down_index=[0, 0, 0], attn_res=[], in_res=7),
}
BATCH = {"l0": 3, "l1": 4}


def build(ref, cfg):
    ref.LatentDiffusion.attn_resolutions = cfg["attn_res"]
    torch.manual_seed(cfg["manual_seed"])
    model = ref.LatentDiffusion(cfg)
    model.curr_device = torch.device("cpu")
    return model


def make(name):
    ref = load_reference_ldm()
    cfg = CONFIGS[name]
    B, R, C0 = BATCH[name], cfg["in_res"], cfg["h_dims"][0]
    out = {}
    model = build(ref, cfg)
    for k, v in model.state_dict().items():
        out["state0/" + k] = v.clone()
    x0 = torch.randn((B, C0, R, R), generator=torch.Generator().manual_seed(7))      # latents: zero mean, unit scale
    y = torch.zeros(B, dtype=torch.int64)
    out["x0"] = x0
    eps, t = _draw(100, x0)
    out["eps"], out["t"] = eps, t
    n, s = model.diffusion_schedule(t)
    out["sched_n"], out["sched_s"] = n, s
    x_t = s * x0 + n * eps
    out["x_t"] = x_t

    model.eval()
    with torch.no_grad():
        out["fwd_eval"] = model(x_t, n ** 2)
    model.train()
    taps, hooks = {}, []
    for sec in ("Downsampling", "Middle", "Upsampling"):
        for j, m in enumerate(getattr(model, sec)):
            hooks.append(m.register_forward_hook(lambda mod, i, o, key=f"{sec}.{j}": taps.__setitem__(key, o.detach().clone())))
    torch.manual_seed(100)
    loss = model.p_loss(x0)
    for h in hooks:
        h.remove()
    for k, v in taps.items():
        out["tap/" + k] = v
    model.zero_grad()
    loss.backward()
    out["loss"] = loss.detach().reshape(1)
    for k, p in model.named_parameters():
        out["grad/" + k] = p.grad.detach().clone()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["bn1/" + k] = v.clone()

    model = build(ref, cfg)
    model.train()
    opts, scheds = model.configure_optimizers()
    opt = opts[0]
    assert type(opt) is torch.optim.AdamW
    losses = []
    for step in range(3):
        torch.manual_seed(100 + step)
        loss = model.training_step((x0, y), step)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    out["loss_steps"] = torch.stack(losses)
    for k, v in model.state_dict().items():
        out["state3/" + k] = v.clone()

    model.eval()
    with torch.no_grad():
        torch.manual_seed(200)
        out["val_loss"] = model.p_loss(x0).reshape(1)
        gen_noise = torch.randn((2, C0, R, R), generator=torch.Generator().manual_seed(11))
        out["gen_noise"] = gen_noise
        out["gen_T4"] = model.generate(2, 4, initial_noise=gen_noise.clone())

    out = {k: v.contiguous() for k, v in out.items()}
    path = os.path.join(OUT_DIR, f"{name}.safetensors")
    save_file(out, path, metadata={"config": json.dumps(cfg), "batch": str(B),
                                   "seeds": json.dumps({"draw_step_k": "100+k", "val": 200, "x0": 7, "gen_noise": 11})})
    nparam = sum(v.numel() for k, v in out.items() if k.startswith("grad/"))
    print(f"{name}: {len(out)} tensors, {nparam} params, loss {float(out['loss']):.5f}, {os.path.getsize(path)/1e6:.2f} MB -> {path}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for nm in (sys.argv[1:] or list(CONFIGS)):
        make(nm)
