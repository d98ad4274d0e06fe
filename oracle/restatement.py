"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference DDPM hot path.

A plain-PyTorch (CPU, fp32 unless told otherwise), *functional* restatement of
`models/ddpm.py` of the reference: the UNet forward, the cosine schedule, the
L1 noise-prediction loss, the deterministic sampler and the AdamW update.  It is
the checker for the HIP path (tests/, `__graft_entry__.smoke()`) and the
`cpu_baseline` leg of `bench.py`; the product package never imports it.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function
here against fixtures under `tests/golden/` that `oracle/make_golden.py`
produced by running the unmodified reference file in the build container (the
reference repository itself ships no tests or golden vectors).

State is a flat ``dict[str, Tensor]`` with exactly the reference's
``state_dict`` keys, so reference checkpoints load unchanged.  Every function
cites the reference lines it follows.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01      # nn.LeakyReLU() default, reference models/ddpm.py:144
BN_EPS = 1e-5           # nn.BatchNorm2d default, models/ddpm.py:50,56
BN_MOMENTUM = 0.1
GN_EPS = 1e-6           # models/ddpm.py:93
GN_GROUPS = 32


# --------------------------------------------------------------------------- plan
def build_plan(cfg) -> Dict[str, List[tuple]]:
    """Module sequence of the UNet, from reference models/ddpm.py:160-204.

    Returns {'down': [...], 'mid': [...], 'up': [...], 'end': (c, c)} where each
    entry is ('res', cin, cout, p_drop) | ('attn', c) | ('down', c) | ('up', c).
    The attention test uses the running feature resolution, which turns into a
    float after the first halving (models/ddpm.py:170-175, :191-198).
    """
    h = list(cfg["h_dims"])
    di = list(cfg["down_index"])
    attn = list(cfg["attn_res"])
    res = cfg["patch_size"]
    down, mid, up = [], [], []
    cout = h[0]
    for i in range(len(h) - 1):
        cin, cout = h[i], h[i + 1]
        down.append(("res", cin, cout, 0.0))
        if res in attn:
            down.append(("attn", cout))
        if di[i + 1] == 1:
            down.append(("down", cout))
            res = res / 2
    p = cfg["dropout"]
    mid = [("res", cout, cout, p), ("attn", cout), ("res", cout, cout, p)]
    rh, rdi = h[::-1], di[::-1]
    for i in range(len(h) - 1):
        cin, cout = rh[i], rh[i + 1]
        if rdi[i] == 1:
            up.append(("up", cin))
            res = res * 2
        up.append(("res", cin, cout, 0.0))
        if res in attn:
            up.append(("attn", cout))
    return {"down": down, "mid": mid, "up": up, "end": (cout, cout),
            "ch": h[1], "with_conv": bool(cfg["with_conv"])}


def state_template(cfg) -> Dict[str, Tuple[tuple, torch.dtype]]:
    """Names, shapes and dtypes of every state_dict entry, in creation order
    (reference models/ddpm.py:156-204 and ResNetBlock/AttnBlock constructors :43-98)."""
    plan = build_plan(cfg)
    out: Dict[str, Tuple[tuple, torch.dtype]] = {}
    f32 = torch.float32

    def conv(name, co, ci, k):
        out[name + ".weight"] = ((co, ci, k, k), f32)
        out[name + ".bias"] = ((co,), f32)

    def bn(name, c):
        out[name + ".weight"] = ((c,), f32)
        out[name + ".bias"] = ((c,), f32)
        out[name + ".running_mean"] = ((c,), f32)
        out[name + ".running_var"] = ((c,), f32)
        out[name + ".num_batches_tracked"] = ((), torch.int64)

    ch = plan["ch"]
    conv("linear_1", 4 * ch, ch, 1)
    conv("linear_2", 4 * ch, 4 * ch, 1)

    def module(prefix, m):
        if m[0] == "res":
            _, ci, co, _ = m
            conv(prefix + ".linear_temb", co, 4 * ch, 1)
            bn(prefix + ".block1.0.0", ci)
            conv(prefix + ".block1.0.2", co, ci, 3)
            bn(prefix + ".block2.0.0", co)
            conv(prefix + ".block2.0.3", co, co, 3)
            if ci != co:
                conv(prefix + ".layers.0", co, ci, 1)
        elif m[0] == "attn":
            c = m[1]
            out[prefix + ".norm.weight"] = ((c,), f32)
            out[prefix + ".norm.bias"] = ((c,), f32)
            for p in ("q_proj", "k_proj", "v_proj", "linear"):
                conv(prefix + "." + p, c, c, 1)
        elif m[0] == "down":
            if plan["with_conv"]:
                conv(prefix + ".layers.0", m[1], m[1], 3)
        elif m[0] == "up":
            if plan["with_conv"]:
                conv(prefix + ".layers.1", m[1], m[1], 3)

    for sec, key in (("Downsampling", "down"), ("Middle", "mid"), ("Upsampling", "up")):
        for j, m in enumerate(plan[key]):
            module(f"{sec}.{j}", m)
    conv("End.0.0", plan["end"][0], plan["end"][1], 1)
    return out


def is_param(name: str) -> bool:
    return not name.endswith(("running_mean", "running_var", "num_batches_tracked"))


# ---------------------------------------------------------------------- primitives
def timeembed(t: torch.Tensor, dim: int) -> torch.Tensor:
    """Sinusoidal embedding of t [B,1,1,1] -> [B,dim,1,1]; reference models/ddpm.py:125-136.

    The angles reach 2*pi*1000*t ~ 6e3 rad, where one float32 ulp is 5e-4 rad: the float32
    rounding of the angle is part of what the reference computes.  The angles are therefore always
    formed in float32, also when the rest of the network is evaluated in float64."""
    out_dtype = t.dtype
    t = t.float()
    f = torch.exp(torch.linspace(math.log(1.0), math.log(1000.0), dim // 2,
                                 device=t.device)).reshape(1, dim // 2, 1, 1)
    w = 2.0 * math.pi * f
    return torch.cat([torch.sin(w * t), torch.cos(w * t)], dim=1).to(out_dtype)


def diffusion_schedule(t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(noise_rate, signal_rate) for diffusion time t in [0,1]; models/ddpm.py:242-255."""
    lo = torch.acos(torch.tensor(0.95, device=t.device))
    hi = torch.acos(torch.tensor(0.02, device=t.device))
    ang = lo + t * (hi - lo)
    return torch.sin(ang), torch.cos(ang)


def _act(x):
    return F.leaky_relu(x, LEAKY_SLOPE)


def _bn(sd, name, x, training):
    y = F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                     sd[name + ".weight"], sd[name + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[name + ".num_batches_tracked"] += 1
    return y


def resnet_block(sd, pfx, x, temb, training, p_drop, drop_mask=None):
    """reference models/ddpm.py:70-87 (block1 :49-53, block2 :55-60, shortcut :63-68)."""
    h = F.conv2d(_act(_bn(sd, pfx + ".block1.0.0", x, training)),
                 sd[pfx + ".block1.0.2.weight"], sd[pfx + ".block1.0.2.bias"], padding=1)
    h = h + F.conv2d(_act(temb), sd[pfx + ".linear_temb.weight"], sd[pfx + ".linear_temb.bias"])
    a = _act(_bn(sd, pfx + ".block2.0.0", h, training))
    if drop_mask is not None:
        a = a * drop_mask                    # explicit mask already scaled by 1/(1-p)
    elif training and p_drop > 0.0:
        a = F.dropout(a, p_drop, True)
    h = F.conv2d(a, sd[pfx + ".block2.0.3.weight"], sd[pfx + ".block2.0.3.bias"], padding=1)
    if pfx + ".layers.0.weight" in sd:
        x = F.conv2d(x, sd[pfx + ".layers.0.weight"], sd[pfx + ".layers.0.bias"])
    return x + h


def attn_block(sd, pfx, x):
    """Single-head spatial self-attention; reference models/ddpm.py:100-123."""
    B, C, H, W = x.shape
    h = F.group_norm(x, GN_GROUPS, sd[pfx + ".norm.weight"], sd[pfx + ".norm.bias"], GN_EPS)
    q = F.conv2d(h, sd[pfx + ".q_proj.weight"], sd[pfx + ".q_proj.bias"]).reshape(B, C, H * W)
    k = F.conv2d(h, sd[pfx + ".k_proj.weight"], sd[pfx + ".k_proj.bias"]).reshape(B, C, H * W)
    v = F.conv2d(h, sd[pfx + ".v_proj.weight"], sd[pfx + ".v_proj.bias"]).reshape(B, C, H * W)
    w = torch.bmm(q.transpose(1, 2), k) * (C ** -0.5)        # [B, S(query), s(key)]
    w = torch.softmax(w, dim=-1)
    o = torch.bmm(v, w.transpose(1, 2)).reshape(B, C, H, W)  # out[c,S] = sum_s v[c,s] w[S,s]
    o = F.conv2d(o, sd[pfx + ".linear.weight"], sd[pfx + ".linear.bias"])
    return x + o


def down_block(sd, pfx, x, with_conv=True):
    """reference models/ddpm.py:14-27."""
    if with_conv:
        return F.conv2d(x, sd[pfx + ".layers.0.weight"], sd[pfx + ".layers.0.bias"],
                        stride=2, padding=1)
    return F.avg_pool2d(x, 2)


def up_block(sd, pfx, x, with_conv=True):
    """reference models/ddpm.py:29-41."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    if with_conv:
        x = F.conv2d(x, sd[pfx + ".layers.1.weight"], sd[pfx + ".layers.1.bias"], padding=1)
    return x


def unet_forward(sd, cfg, x, t, training=False, drop_masks=None, taps=None):
    """reference models/ddpm.py:208-239.  `t` is [B,1,1,1] (the reference passes
    noise_rate**2).  `drop_masks` maps a module prefix to an explicit, pre-scaled
    dropout mask.  `taps`, if a dict, receives every module output by prefix."""
    plan = build_plan(cfg)
    wc = plan["with_conv"]
    drop_masks = drop_masks or {}
    temb = timeembed(t, plan["ch"])
    temb = F.conv2d(temb, sd["linear_1.weight"], sd["linear_1.bias"])
    temb = F.conv2d(_act(temb), sd["linear_2.weight"], sd["linear_2.bias"])
    if taps is not None:
        taps["temb"] = temb

    def run(pfx, m, h):
        if m[0] == "res":
            h = resnet_block(sd, pfx, h, temb, training, m[3], drop_masks.get(pfx))
        elif m[0] == "attn":
            h = attn_block(sd, pfx, h)
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
            h = h + skip[j]                    # `h += skip[index]`, models/ddpm.py:230-231
        h = run(f"Upsampling.{j}", m, h)
    return F.conv2d(h, sd["End.0.0.weight"], sd["End.0.0.bias"])


# ------------------------------------------------------------------- loss / sampler
def p_loss(sd, cfg, images, noises, diffusion_times, training=True, drop_masks=None, taps=None):
    """L1 noise-prediction loss; reference models/ddpm.py:262-269 with the two random
    draws (`randn_like` :263, `rand` :264) supplied by the caller."""
    n, s = diffusion_schedule(diffusion_times.float())       # rates are float32 in the reference
    t_net = (n ** 2).to(images.dtype)                        # network conditioning, :258
    n, s = n.to(images.dtype), s.to(images.dtype)
    noisy = s * images + n * noises
    pred = unet_forward(sd, cfg, noisy, t_net, training, drop_masks, taps)    # denoise, :257-258
    if taps is not None:
        taps["pred_noises"] = pred
    return F.l1_loss(pred, noises)


def reverse_diffusion(sd, cfg, initial_noise, steps):
    """Deterministic sampler; reference models/ddpm.py:271-288 (returns the last pred_images)."""
    B = initial_noise.shape[0]
    dt = 1.0 / steps
    cur = initial_noise
    pred_images = None
    for k in range(steps):
        t = torch.ones((B, 1, 1, 1), device=cur.device) - k * dt
        n, s = diffusion_schedule(t)
        eps = unet_forward(sd, cfg, cur, n ** 2, training=False)
        pred_images = (cur - n * eps) / s                       # :259
        n2, s2 = diffusion_schedule(t - dt)
        cur = s2 * pred_images + n2 * eps                       # :285-287
    return pred_images


# ---------------------------------------------------------------------- optimiser
def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, weight_decay,
               beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW as configured at reference models/ddpm.py:340-345
    (decoupled weight decay, bias-corrected moments).  In place; `step` is the
    1-based step number."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k in params:
        p, g = params[k], grads[k]
        p.mul_(1.0 - lr * weight_decay)
        exp_avg[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        exp_avg_sq[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (exp_avg_sq[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(exp_avg[k], denom, value=-lr / bc1)


def loss_and_grads(sd, cfg, images, noises, diffusion_times, drop_masks=None):
    """Training-mode p_loss and d loss / d parameter for every parameter tensor.
    BN running statistics in `sd` are updated as a side effect (train mode)."""
    leaves = {}
    work = {}
    for k, v in sd.items():
        if is_param(k):
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
        else:
            work[k] = v
    loss = p_loss(work, cfg, images, noises, diffusion_times, True, drop_masks)
    names = list(leaves)
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(names, gs)}
    return loss.detach(), grads


def init_state_like_torch(cfg, seed):
    """Fresh state with torch's default Conv2d / BatchNorm2d / GroupNorm initialisation
    drawn in the reference's construction order (run.py:44-46 seeds, then builds).
    Only used to make synthetic weights for benchmarks; parity tests use the
    committed golden state."""
    torch.manual_seed(seed)
    sd = {}
    tmpl = state_template(cfg)
    names = list(tmpl)
    i = 0
    while i < len(names):
        k = names[i]
        shape, dt = tmpl[k]
        if k.endswith(".weight") and len(shape) == 4:
            w = torch.empty(shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            b = torch.empty(shape[0]).uniform_(-bound, bound)
            sd[k] = w
            sd[names[i + 1]] = b
            i += 2
            continue
        if k.endswith("running_var") or (k.endswith(".weight") and len(shape) == 1):
            sd[k] = torch.ones(shape, dtype=dt)
        else:
            sd[k] = torch.zeros(shape, dtype=dt)
        i += 1
    return sd
