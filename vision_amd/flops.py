"""Algorithmic FLOP counts (2 x multiply-adds) of the UNet, from the layer plan."""
from .plan import make_plan


def conv_flops(cin, cout, k, ho, wo):
    return 2.0 * cin * cout * k * k * ho * wo


def forward_flops_per_image(config):
    """(total, first_layer_dgrad) where first_layer_dgrad is the part of the input gradient that
    is never needed (the two convs that read the network input)."""
    plan = make_plan(config)
    res = config["patch_size"]
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
