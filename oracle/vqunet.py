"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference VQ-VAE-UNet first stage.

A plain-PyTorch (CPU, fp32), functional restatement of `models/vqvaeunet.py` of the reference: encoder, codebook
lookup, decoder with the additive skips, sigmoid head, reconstruction + codebook loss, and the Adam update with L2
weight decay.  It is the checker for the HIP path in tests/; the product package never imports it.

Parity status: PINNED.  `tests/test_vqunet_cpu.py` checks every function here against `tests/golden/vqunet.safetensors`,
which `oracle/make_golden_vqunet.py` produced by running the unmodified reference file in the build container.

State is a flat ``dict[str, Tensor]`` with the reference's ``state_dict`` keys.
"""
from typing import Dict, List

import torch
import torch.nn.functional as F

from oracle.restatement import _act, _bn, down_block, up_block

BETA = 0.25             # VQBlock default, reference models/vqvaeunet.py:106


def build_plan(cfg) -> Dict[str, List[tuple]]:
    """Module sequence, from reference models/vqvaeunet.py:164-203: ('res', cin, cout) | ('attn', c) | ('down', c) |
    ('up', c)."""
    h = list(cfg["h_dims"])
    di = list(cfg["down_index"])
    attn = list(cfg["attn_res"])
    res = cfg["patch_size"]
    down, up = [], []
    cout = h[0]
    for i in range(len(h) - 1):
        cin, cout = h[i], h[i + 1]
        down.append(("res", cin, cout))
        if res in attn:
            down.append(("attn", cout))
        if di[i + 1] == 1:
            down.append(("down", cout))
            res = res / 2
    rh, rdi = h[::-1], di[::-1]
    for i in range(len(h) - 1):
        cin, cout = rh[i], rh[i + 1]
        if rdi[i] == 1:
            up.append(("up", cin))
            res = res * 2
        up.append(("res", cin, cout))
        if res in attn:
            up.append(("attn", cout))
    return {"down": down, "up": up, "end": cout}


def is_param(name: str) -> bool:
    return not name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def channel_linear(sd, name, x):
    """reference models/Custumnn.py:9-21: nn.Linear over the channel axis of an NCHW tensor."""
    return F.linear(x.permute(0, 2, 3, 1), sd[name + ".linear.weight"], sd[name + ".linear.bias"]).permute(0, 3, 1, 2)


def resnet_block(sd, pfx, x, training):
    """reference models/vqvaeunet.py:69-81 (block1 :49-53, block2 :55-60 with dropout 0, shortcut :63-68)."""
    h = F.conv2d(_act(_bn(sd, pfx + ".block1.0.0", x, training)),
                 sd[pfx + ".block1.0.2.weight"], sd[pfx + ".block1.0.2.bias"], padding=1)
    h = F.conv2d(_act(_bn(sd, pfx + ".block2.0.0", h, training)),
                 sd[pfx + ".block2.0.3.weight"], sd[pfx + ".block2.0.3.bias"], padding=1)
    if pfx + ".layers.0.linear.weight" in sd:
        x = channel_linear(sd, pfx + ".layers.0", x)
    return x + h


def attn_block(sd, pfx, x, training):
    """Attention between channels; reference models/vqvaeunet.py:90-100: one projection serves as q, k and v."""
    B, C, H, W = x.shape
    h = _bn(sd, pfx + ".norm", x, training)
    p = channel_linear(sd, pfx + ".linear1", h).reshape(B, C, H * W)
    w = torch.softmax(torch.matmul(p, p.transpose(-2, -1)) * (C ** -0.5), dim=-1)
    return x + channel_linear(sd, pfx + ".linear2", torch.matmul(w, p).reshape(B, C, H, W))


def vq_block(weight, latents, beta=BETA):
    """reference models/vqvaeunet.py:115-146, same torch operations in the same order.  Returns (quantised latents
    with the straight-through gradient, vq_loss, codes [B*H*W])."""
    lat = latents.permute(0, 2, 3, 1).contiguous()
    flat = lat.view(-1, weight.shape[1])
    dist = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(weight ** 2, dim=1) - 2 * torch.matmul(flat, weight.t())
    idx = torch.argmin(dist, dim=1)
    one_hot = torch.zeros(idx.numel(), weight.shape[0])
    one_hot.scatter_(1, idx.unsqueeze(1), 1)
    q = torch.matmul(one_hot, weight).view(lat.shape)
    vq_loss = F.mse_loss(q.detach(), lat) * beta + F.mse_loss(q, lat.detach())
    q = lat + (q - lat).detach()
    return q.permute(0, 3, 1, 2).contiguous(), vq_loss, idx


def _run(sd, pfx, m, h, training, taps):
    if m[0] == "res":
        h = resnet_block(sd, pfx, h, training)
    elif m[0] == "attn":
        h = attn_block(sd, pfx, h, training)
    elif m[0] == "down":
        h = down_block(sd, pfx, h)
    else:
        h = up_block(sd, pfx, h)
    if taps is not None:
        taps[pfx] = h
    return h


def encode(sd, cfg, x, training=False, taps=None):
    """reference models/vqvaeunet.py:215-222: (encoder output, every encoder module's output)."""
    h = x
    skip = []
    for j, m in enumerate(build_plan(cfg)["down"]):
        h = _run(sd, f"Downsampling.{j}", m, h, training, taps)
        skip.append(h)
    return h, skip


def forward(sd, cfg, x, training=False, taps=None):
    """reference models/vqvaeunet.py:205-233: (x_rec, vq_loss, codes).  `taps`, if a dict, receives every module
    output by prefix, plus 'latents' and 'quantized'."""
    plan = build_plan(cfg)
    h, skip = encode(sd, cfg, x, training, taps)
    q, vq_loss, idx = vq_block(sd["vq_layer.embedding.weight"], h)
    if taps is not None:
        taps["latents"], taps["quantized"] = h, q
    h = q
    skip.reverse()
    n_up = len(plan["up"])
    for j, m in enumerate(plan["up"]):
        if j < len(skip):
            h = h + skip[j]                    # `h += skip[index]`, models/vqvaeunet.py:229-230
        h = _run(sd, f"Upsampling.{j}", m, h, training, taps)
    if n_up < len(skip):
        h = h + skip[n_up]
    h = torch.sigmoid(channel_linear(sd, f"Upsampling.{n_up}.0", h))
    return h, vq_loss, idx


def loss(sd, cfg, x, training=True, taps=None):
    """reference models/vqvaeunet.py:235-247: mse(x_rec, x) + vq_loss; also returns the parts."""
    x_rec, vq_loss, idx = forward(sd, cfg, x, training, taps)
    rec = F.mse_loss(x_rec, x)
    return rec + vq_loss, rec, vq_loss, x_rec, idx


def loss_and_grads(sd, cfg, x):
    """Training-mode loss and its gradient for every parameter; BN running statistics in `sd` are updated."""
    leaves, work = {}, {}
    for k, v in sd.items():
        if is_param(k):
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
        else:
            work[k] = v
    total, rec, vq_loss, x_rec, idx = loss(work, cfg, x, True)
    names = list(leaves)
    gs = torch.autograd.grad(total, [leaves[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(names, gs)}
    return total.detach(), rec.detach(), vq_loss.detach(), x_rec.detach(), idx, grads


def adam_l2_step(params, grads, exp_avg, exp_avg_sq, step, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam as configured at reference models/vqvaeunet.py:263-269: the L2 term joins the gradient before
    the moment updates.  In place; `step` is the 1-based step number."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k in params:
        p = params[k]
        g = grads[k].add(p, alpha=weight_decay)
        exp_avg[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        exp_avg_sq[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (exp_avg_sq[k].sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(exp_avg[k], denom, value=-lr / bc1)
