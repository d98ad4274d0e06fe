"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference latent-diffusion network and objective.

A plain-PyTorch (CPU, fp32 or fp64), functional restatement of `models/ldm.py` of the reference: the UNet over latent
maps (blocks with channel-wise Linear projections, the channel-attention block), the L1 noise-prediction loss and the
deterministic sampler.  The schedule, the time embedding and AdamW are the DDPM ones (models/ldm.py:111-122, :246-258,
:346-352 repeat models/ddpm.py) and come from oracle/restatement.py.  Checker for the HIP path in tests/ only.

Parity status: PINNED on `tests/golden/{l0,l1}.safetensors`, which `oracle/make_golden_ldm.py` produced by running the
reference's own `LatentDiffusion.forward` / `p_loss` / `generate` (the class is made constructible by supplying, from
outside, the one attribute its constructor reads and never sets; the file is not modified).

State is a flat ``dict[str, Tensor]`` with the reference's ``state_dict`` keys.
"""
import torch
import torch.nn.functional as F

from oracle.restatement import _act, _bn, diffusion_schedule, down_block, timeembed, up_block
from oracle.restatement import build_plan as _ddpm_plan
from oracle.vqunet import attn_block, channel_linear, is_param  # noqa: F401  (the attention block of models/ldm.py:91-108 is
#                                                                  the one of models/vqvaeunet.py:83-100; is_param for the tests)


def build_plan(cfg):
    """reference models/ldm.py:170-221: the DDPM module sequence, resolution counted from `in_res`."""
    return _ddpm_plan(dict(cfg, patch_size=cfg["in_res"]))


def resnet_block(sd, pfx, x, temb, training, p_drop, drop_mask=None):
    """reference models/ldm.py:71-89."""
    h = F.conv2d(_act(_bn(sd, pfx + ".block1.0.0", x, training)),
                 sd[pfx + ".block1.0.2.weight"], sd[pfx + ".block1.0.2.bias"], padding=1)
    h = h + channel_linear(sd, pfx + ".linear_temb", _act(temb))
    a = _act(_bn(sd, pfx + ".block2.0.0", h, training))
    if drop_mask is not None:
        a = a * drop_mask
    elif training and p_drop > 0.0:
        a = F.dropout(a, p_drop, True)
    h = F.conv2d(a, sd[pfx + ".block2.0.3.weight"], sd[pfx + ".block2.0.3.bias"], padding=1)
    if pfx + ".layers.0.linear.weight" in sd:
        x = channel_linear(sd, pfx + ".layers.0", x)
    return x + h


def unet_forward(sd, cfg, x, t, training=False, drop_masks=None, taps=None):
    """reference models/ldm.py:225-244."""
    plan = build_plan(cfg)
    wc = plan["with_conv"]
    drop_masks = drop_masks or {}
    temb = channel_linear(sd, "linear_1", timeembed(t, plan["ch"]))
    temb = channel_linear(sd, "linear_2", _act(temb))

    def run(pfx, m, h):
        if m[0] == "res":
            h = resnet_block(sd, pfx, h, temb, training, m[3], drop_masks.get(pfx))
        elif m[0] == "attn":
            h = attn_block(sd, pfx, h, training)
        elif m[0] == "down":
            h = down_block(sd, pfx, h, wc)
        else:
            h = up_block(sd, pfx, h, wc)
        if taps is not None:
            taps[pfx] = h
        return h

    h = x
    skip = []
    for j, m in enumerate(plan["down"]):
        h = run(f"Downsampling.{j}", m, h)
        skip.append(h)
    for j, m in enumerate(plan["mid"]):
        h = run(f"Middle.{j}", m, h)
    skip.reverse()
    for j, m in enumerate(plan["up"]):
        if j < len(skip):
            h = h + skip[j]
        h = run(f"Upsampling.{j}", m, h)
    return channel_linear(sd, "End.0.0", h)


def p_loss(sd, cfg, latents, noises, diffusion_times, training=True, drop_masks=None, taps=None):
    """reference models/ldm.py:264-271 with the two random draws supplied by the caller."""
    n, s = diffusion_schedule(diffusion_times.float())
    t_net = (n ** 2).to(latents.dtype)
    n, s = n.to(latents.dtype), s.to(latents.dtype)
    pred = unet_forward(sd, cfg, s * latents + n * noises, t_net, training, drop_masks, taps)
    return F.l1_loss(pred, noises)


def reverse_diffusion(sd, cfg, initial_noise, steps):
    """reference models/ldm.py:273-292."""
    B = initial_noise.shape[0]
    dt = 1.0 / steps
    cur = initial_noise
    pred_images = None
    for k in range(steps):
        t = torch.ones((B, 1, 1, 1)) - k * dt
        n, s = diffusion_schedule(t)
        eps = unet_forward(sd, cfg, cur, n ** 2, training=False)
        pred_images = (cur - n * eps) / s
        n2, s2 = diffusion_schedule(t - dt)
        cur = s2 * pred_images + n2 * eps
    return pred_images


def loss_and_grads(sd, cfg, latents, noises, diffusion_times, dtype=None):
    """Training-mode p_loss and its gradient for every parameter (in `dtype` when given, e.g. float64)."""
    cast = (lambda v: v.to(dtype)) if dtype is not None else (lambda v: v)
    leaves = {k: cast(v).detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    work = {k: (cast(v).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    work.update(leaves)
    loss = p_loss(work, cfg, cast(latents), cast(noises), diffusion_times, True)
    gs = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(leaves, gs)}
    return loss.detach(), grads
