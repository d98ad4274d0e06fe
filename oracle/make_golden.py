"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/*.safetensors.

Runs the UNMODIFIED reference `models/ddpm.py` (through oracle/ref_loader.py) on
tiny configurations and records inputs and outputs of every function on the hot
path.  Only runs where the reference is mounted (the build container):

    python -m oracle.make_golden

The fixtures are data only (tensors + the config as JSON metadata).
"""
import json
import os
import sys

import torch
import torch.nn.functional as F
from safetensors.torch import save_file

from oracle.ref_loader import load_reference_ddpm

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

BASE = dict(name="ddpm", input_channels=3, learning_rate=1e-3, manual_seed=42, with_conv=True,
            sampling_period=1, weight_decay=1e-2, diffusion_steps=4, scheduler_gamma=0.99, grid=2)

CONFIGS = {
    # Middle + Down-path + Up-path attention, 1 channel per GroupNorm group, taps recorded.
    "t0": dict(BASE, h_dims=[3, 32, 32], down_index=[0, 1, 1], attn_res=[8], patch_size=16,
               dropout=0.0, latent_dim=[3, 16, 16]),
    # flowers-like wiring: channel growth per level, in != out residual blocks both ways,
    # attention in Down / Middle / Up (the ffhq arrangement), odd batch.
    "t1": dict(BASE, h_dims=[3, 32, 64, 64], down_index=[0, 1, 1, 1], attn_res=[8, 4], patch_size=32,
               dropout=0.0, latent_dim=[3, 32, 32]),
    # t0 with dropout 0.1 in the two Middle blocks, masks recorded.
    "t0_drop": dict(BASE, h_dims=[3, 32, 32], down_index=[0, 1, 1], attn_res=[8], patch_size=16,
                    dropout=0.1, latent_dim=[3, 16, 16]),
    # t0 built without resampling convolutions: AvgPool2d((2, 2)) down, bare nearest Upsample up
    # (reference models/ddpm.py:21,33).
    "t0_pool": dict(BASE, h_dims=[3, 32, 32], down_index=[0, 1, 1], attn_res=[8], patch_size=16,
                    dropout=0.0, latent_dim=[3, 16, 16], with_conv=False),
}
BATCH = {"t0": 4, "t1": 3, "t0_drop": 4, "t0_pool": 4}


class _DropoutRecorder:
    """Replaces torch.nn.functional.dropout while the reference runs so that the
    Bernoulli masks it applies (nn.Dropout, reference models/ddpm.py:58) are known."""

    def __init__(self):
        self.masks = []
        self._orig = None

    def __enter__(self):
        self._orig = F.dropout

        def patched(input, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return input
            mask = torch.bernoulli(torch.full_like(input, 1.0 - p)) / (1.0 - p)
            self.masks.append(mask)
            return input * mask

        F.dropout = patched
        return self

    def __exit__(self, *exc):
        F.dropout = self._orig


def _draw(seed, images):
    """The two draws of reference p_loss (models/ddpm.py:263-264) for a given seed."""
    torch.manual_seed(seed)
    eps = torch.randn_like(images)
    t = torch.rand((images.shape[0], 1, 1, 1))
    return eps, t


def make(name):
    ref = load_reference_ddpm()
    cfg = CONFIGS[name]
    B, P = BATCH[name], cfg["patch_size"]
    out = {}

    torch.manual_seed(cfg["manual_seed"])           # run.py:44 seed_everything
    model = ref.DDPMModel(cfg)                       # run.py:46
    for k, v in model.state_dict().items():
        out["state0/" + k] = v.clone()

    g = torch.Generator().manual_seed(7)
    x0 = torch.rand((B, 3, P, P), generator=g)       # ImageFolder-shaped: float32 in [0,1)
    y = torch.zeros(B, dtype=torch.int64)
    out["x0"] = x0
    eps, t = _draw(100, x0)
    out["eps"], out["t"] = eps, t

    n, s = model.diffusion_schedule(t)
    out["sched_n"], out["sched_s"] = n, s
    out["timeembed"] = ref.timeembed(n ** 2, model.ch)
    x_t = s * x0 + n * eps
    out["x_t"] = x_t

    # forward, eval mode, at the initial state
    model.eval()
    with torch.no_grad():
        out["fwd_eval"] = model(x_t, n ** 2)
        pn, pi = model.denoise(x_t, n, s)
        out["denoise_pred_images"] = pi
    # forward + loss + grads, train mode, first step
    model.train()
    taps = {}
    hooks = []
    if name == "t0":
        for sec in ("Downsampling", "Middle", "Upsampling"):
            for j, m in enumerate(getattr(model, sec)):
                hooks.append(m.register_forward_hook(
                    lambda mod, i, o, key=f"{sec}.{j}": taps.__setitem__(key, o.detach().clone())))
    with _DropoutRecorder() as rec:
        torch.manual_seed(100)
        loss = model.p_loss(x0)
    for h in hooks:
        h.remove()
    for k, v in taps.items():
        out["tap/" + k] = v
    for i, m in enumerate(rec.masks):
        out[f"dropmask/{i}"] = m
    model.zero_grad()
    loss.backward()
    out["loss"] = loss.detach().reshape(1)
    for k, p in model.named_parameters():
        out["grad/" + k] = p.grad.detach().clone()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["bn1/" + k] = v.clone()

    # three optimiser steps from the initial state (AdamW from configure_optimizers)
    torch.manual_seed(cfg["manual_seed"])
    model = ref.DDPMModel(cfg)
    model.train()
    opts, scheds = model.configure_optimizers()
    opt = opts[0]
    losses = []
    for step in range(3):
        with _DropoutRecorder() as rec:
            torch.manual_seed(100 + step)
            loss = model.training_step((x0, y), step)
        for i, m in enumerate(rec.masks):
            out[f"step{step}/dropmask/{i}"] = m
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    out["loss_steps"] = torch.stack(losses)
    for k, v in model.state_dict().items():
        out["state3/" + k] = v.clone()

    # validation-style loss and fixed-noise sampling in eval mode from the trained state
    model.eval()
    with torch.no_grad():
        torch.manual_seed(200)
        out["val_loss"] = model.p_loss(x0).reshape(1)
        gen_noise = torch.randn((2, 3, P, P), generator=torch.Generator().manual_seed(11))
        out["gen_noise"] = gen_noise
        out["gen_T4"] = model.generate(2, 4, initial_noise=gen_noise.clone())
        out["gen_T1"] = model.generate(2, 1, initial_noise=gen_noise.clone())

    out = {k: v.contiguous() for k, v in out.items()}
    os.makedirs(OUT_DIR, exist_ok=True)
    path = os.path.join(OUT_DIR, f"{name}.safetensors")
    save_file(out, path, metadata={"config": json.dumps(cfg), "batch": str(B),
                                   "seeds": json.dumps({"draw_step_k": "100+k", "val": 200,
                                                        "x0": 7, "gen_noise": 11})})
    nparam = sum(v.numel() for k, v in out.items() if k.startswith("grad/"))
    print(f"{name}: {len(out)} tensors, {nparam} params, {os.path.getsize(path)/1e6:.2f} MB -> {path}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for nm in (sys.argv[1:] or list(CONFIGS)):
        make(nm)
