"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/vqunet.safetensors.

Runs the UNMODIFIED reference `VQVAEUNETModel` (models/vqvaeunet.py:149-327, loaded by oracle/ref_loader.py) on two
tiny configurations and records, per configuration: the state before the step, the images, a training-mode forward
(every module output, latents, codes, x_rec, both loss terms), the gradient of the total loss for every parameter,
the state after one optimiser step (torch.optim.Adam as `configure_optimizers` builds it), and an eval-mode forward
of the updated model on a second batch.  Only runs where the reference is mounted (the build container):

    python -m oracle.make_golden_vqunet

The codebook is an input like any other weight.  The reference's own initial codebook (uniform +-1/K) puts every
code within fp32 rounding of |z|^2 of every other, so its codes depend on torch's matmul blocking (covered, by
distance, in tests/golden/vq.safetensors); here most codes are placed near encoder outputs, as after training, and
the generator asserts that the best two distances of every latent are well apart in both recorded forwards.
"""
import json
import os

import torch
from safetensors.torch import save_file

from oracle.make_golden_vq import _ArgminRecorder
from oracle.ref_loader import load_reference_vqblock

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vqunet.safetensors")

BASE = dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, sampling_period=1, attn_res=[], weight_decay=1e-3,
            scheduler_gamma=0.99)
CONFIGS = {
    # a level without resampling in the middle (the butterfly arrangement): in == out and in != out blocks both ways
    "v0": dict(BASE, h_dims=[3, 32, 32, 64], down_index=[0, 1, 0, 1], patch_size=16, num_embed=64, embed_dim=64),
    # resampling at every level, odd batch, an AttnBlock on each side at the one level whose block keeps the channel count
    # (at any other level the reference's `h += skip[index]` meets tensors of different widths and fails)
    "v1": dict(BASE, h_dims=[3, 32, 32, 64], down_index=[0, 1, 1, 1], patch_size=16, num_embed=48, embed_dim=64, attn_res=[8]),
}
BATCH = {"v0": 4, "v1": 3}


def _taps(model, store):
    hooks = []
    for sec in ("Downsampling", "Upsampling"):
        for i, m in enumerate(getattr(model, sec)):
            # cloned: the reference adds the next skip into this very tensor (`h += skip[index]`, :230)
            hooks.append(m.register_forward_hook(lambda mod, a, out, key=f"{sec}.{i}": store.__setitem__(key, out.detach().clone())))
    return hooks


def _gap(z_nchw, E):
    z = z_nchw.permute(0, 2, 3, 1).reshape(-1, E.shape[1]).double()
    d = ((z[:, None, :] - E.double()[None]) ** 2).sum(-1)
    two = d.topk(2, dim=1, largest=False).values
    return float(((two[:, 1] - two[:, 0]) / two[:, 1]).min())


def make(ref, name):
    cfg = CONFIGS[name]
    B = BATCH[name]
    torch.manual_seed(cfg["manual_seed"])
    model = ref.VQVAEUNETModel(cfg)
    g = torch.Generator().manual_seed(11)
    x = torch.rand((B, 3, cfg["patch_size"], cfg["patch_size"]), generator=g)
    x2 = torch.rand((B, 3, cfg["patch_size"], cfg["patch_size"]), generator=g)

    # codes near encoder outputs (see the module docstring); the probe forward runs on a copy of the buffers
    model.train()
    keep = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        z, _ = model.encode(x)
    model.load_state_dict(keep)
    K, D = cfg["num_embed"], cfg["embed_dim"]
    zf = z.permute(0, 2, 3, 1).reshape(-1, D)
    sigma = float(zf.std())
    centres = torch.randperm(zf.shape[0], generator=g)[: (5 * K) // 8]
    E = torch.randn((K, D), generator=g) * sigma
    E[: centres.numel()] = zf[centres] + 0.1 * sigma * torch.randn((centres.numel(), D), generator=g)
    with torch.no_grad():
        model.vq_layer.embedding.weight.copy_(E)

    out = {f"state/{k}": v.clone() for k, v in model.state_dict().items()}
    out["x"], out["x2"] = x, x2

    taps = {}
    hooks = _taps(model, taps)
    with _ArgminRecorder() as rec:
        x_rec, vq_loss = model(x)
    for h in hooks:
        h.remove()
    loss = model.loss_function(x, x_rec, vq_loss)
    model.zero_grad()
    loss.backward()
    gap_train = _gap(taps[f"Downsampling.{len(model.Downsampling) - 1}"], E)
    out.update({f"train/tap/{k}": v for k, v in taps.items()})
    out.update({"train/x_rec": x_rec.detach(), "train/vq_loss": vq_loss.detach().reshape(1),
                "train/loss": loss.detach().reshape(1), "train/indices": rec.seen[0].view(-1)})
    out.update({f"grad/{k}": p.grad.clone() for k, p in model.named_parameters()})

    opt = model.configure_optimizers()[0][0]
    assert type(opt) is torch.optim.Adam
    opt.step()
    out.update({f"after/{k}": v.clone() for k, v in model.state_dict().items()})

    model.eval()
    taps = {}
    hooks = _taps(model, taps)
    with torch.no_grad(), _ArgminRecorder() as rec:
        x_rec, vq_loss = model(x2)
        loss = model.loss_function(x2, x_rec, vq_loss)
    for h in hooks:
        h.remove()
    gap_eval = _gap(taps[f"Downsampling.{len(model.Downsampling) - 1}"], model.vq_layer.embedding.weight.detach())
    out.update({"eval/x_rec": x_rec, "eval/vq_loss": vq_loss.reshape(1), "eval/loss": loss.reshape(1),
                "eval/indices": rec.seen[0].view(-1), "eval/latents": taps[f"Downsampling.{len(model.Downsampling) - 1}"]})
    used = out["train/indices"].unique().numel()
    print(f"{name}: loss {float(out['train/loss']):.6f} vq {float(out['train/vq_loss']):.6f}; {used} of {K} codes used; "
          f"relative gap of the best two distances: train {gap_train:.2e}, eval {gap_eval:.2e}")
    assert gap_train > 1e-4 and gap_eval > 1e-4, "codes too close to call in fp32: change the generator seed"
    return {f"{name}/{k}": v.contiguous() for k, v in out.items()}


def main():
    ref = load_reference_vqblock()
    out = {}
    for name in CONFIGS:
        out.update(make(ref, name))
    save_file(out, OUT, metadata={"configs": json.dumps(CONFIGS), "batch": json.dumps(BATCH)})
    print(f"{len(out)} tensors, {os.path.getsize(OUT) / 1e6:.2f} MB -> {OUT}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
