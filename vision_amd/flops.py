"""Algorithmic FLOP counts (2 x multiply-adds) of the UNets, from their layer plans."""
from .plan import make_ldm_plan, make_plan, make_vq_plan


def conv_flops(cin, cout, k, ho, wo):
    return 2.0 * cin * cout * k * k * ho * wo


def forward_flops_per_image(config, plan=None, channel_attention=False):
    """(total, first_layer_dgrad) where first_layer_dgrad is the part of the input gradient that
    is never needed (the two convs that read the network input).  Default: the DDPM plan of `config`."""
    plan = plan or make_plan(config)
    res = plan.patch_size
    total, skip = 0.0, 0.0
    first = True
    for spec in plan.layers:
        if spec.kind == "res":
            f1 = conv_flops(spec.cin, spec.cout, 3, res, res)
            total += f1 + conv_flops(spec.cout, spec.cout, 3, res, res)
            sc = conv_flops(spec.cin, spec.cout, 1, res, res) if spec.cin != spec.cout else 0.0
            total += sc
            if first:
                skip = f1 + sc
                first = False
        elif spec.kind == "attn" and channel_attention:
            # one shared projection + the output projection, scores [C, C] over the positions and their product with p
            s = res * res
            total += 2 * conv_flops(spec.cout, spec.cout, 1, res, res) + 2 * (2.0 * spec.cout * spec.cout * s)
        elif spec.kind == "attn":
            s = res * res
            total += 4 * conv_flops(spec.cout, spec.cout, 1, res, res) + 2 * (2.0 * s * s * spec.cout)
        elif spec.kind == "down":
            res = (res - 1) // 2 + 1 if plan.with_conv else res // 2
            total += conv_flops(spec.cin, spec.cout, 3, res, res) if plan.with_conv else 0.0   # AvgPool2d otherwise
        else:
            res = res * 2
            total += conv_flops(spec.cin, spec.cout, 3, res, res) if plan.with_conv else 0.0   # bare Upsample otherwise
    total += conv_flops(plan.end_channels, plan.end_channels, 1, res, res)
    return total, skip


def train_flops_per_image(config):
    """forward + input gradient + weight gradient = 3 x forward minus the unneeded input gradient."""
    total, skip = forward_flops_per_image(config)
    return 3.0 * total - skip


def first_stage_train_flops_per_image(config):
    """VQ-VAE-UNet (reference models/vqvaeunet.py): convolutions and channel-wise linears as above, plus the codebook
    distances 2 K D per latent vector (forward only: the lookup's gradients are gathers)."""
    plan = make_vq_plan(config)
    total, skip = forward_flops_per_image(config, plan, channel_attention=True)
    latents = plan.section("Downsampling")[-1].res_out ** 2
    return 3.0 * total - skip + 2.0 * config["num_embed"] * config["embed_dim"] * latents


def latent_train_flops_per_image(config):
    """The latent-space UNet of reference models/ldm.py on one latent map."""
    total, skip = forward_flops_per_image(config, make_ldm_plan(config), channel_attention=True)
    return 3.0 * total - skip


def norm_elems_per_image(config, plan=None):
    """(BatchNorm'd, GroupNorm'd) activation elements of one image's forward: every ResNetBlock normalises its input
    (cin channels) and the output of its first convolution (cout channels) at the block's resolution; an attention block
    normalises its input once.  The byte counts of the normalisation passes are multiples of these."""
    plan = plan or make_plan(config)
    res = plan.patch_size
    bn, gn = 0.0, 0.0
    for spec in plan.layers:
        if spec.kind == "res":
            bn += (spec.cin + spec.cout) * res * res
        elif spec.kind == "attn":
            gn += spec.cout * res * res
        elif spec.kind == "down":
            res = (res - 1) // 2 + 1 if plan.with_conv else res // 2
        else:
            res = res * 2
    return bn, gn
