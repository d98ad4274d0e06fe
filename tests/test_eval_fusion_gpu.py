"""Eval-mode forward with the BatchNorm + LeakyReLU passes folded into the producing convolutions (the sampler's
path; reference models/ddpm.py:49-60,74-87 in eval mode) against the same forward with separate passes: the kernels
apply the normalisation to the rounded convolution output, so both are bit-identical."""
import pytest
import torch

import vision_amd
from vision_amd import engine as E
from vision_amd.presets import get_preset
from conftest import load_golden, sub

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _both(model, x, t, monkeypatch):
    out = {}
    for fuse in (True, False):
        monkeypatch.setattr(E, "FUSE_EVAL_NORM", fuse)
        with torch.no_grad():
            out[fuse] = model(x, t).clone()
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("name", ["t0", "t1", "t0_pool"])
def test_golden_configs(name, dtype, monkeypatch):
    g, cfg, B = load_golden(name)
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=dtype)
    m.load_state_dict(sub(g, "state3"))
    m = m.to(DEV).eval()
    out = _both(m, g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV), monkeypatch)
    assert torch.equal(out[True], out[False])
    calls = []
    orig = E.call
    monkeypatch.setattr(E, "call", lambda n, *a: (calls.append(n), orig(n, *a))[1])
    monkeypatch.setattr(E, "FUSE_EVAL_NORM", True)
    with torch.no_grad():
        m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))
    assert calls.count("vamd_conv2d_fwd_act") >= sum(1 for l in m._ensure_engine().layers if isinstance(l, E._ResBlock)) - 1
    # with a tape (gradients in eval mode) nothing is folded: backward needs the tensors in between
    calls.clear()
    x = g["x_t"].to(DEV)
    m(x, (g["sched_n"] ** 2).to(DEV)).sum().backward()
    assert "vamd_conv2d_fwd_act" not in calls


@pytest.mark.parametrize("preset,size,batch", [("butterfly", 224, 2), ("butterfly", 224, 16), ("ffhq", 256, 2)])
def test_full_width_models(preset, size, batch, monkeypatch):
    """The shipped widths (64 ... 512 channels: every kernel family) at 224 x 224 and the ffhq widths at a quarter of
    its resolution; the sampler end to end at 16 images (the split reduction of the 14 x 14 layers included)."""
    cfg = dict(get_preset(preset)["model_config"])
    if size != cfg["patch_size"]:
        f = cfg["patch_size"] // size
        cfg.update(patch_size=size, attn_res=[r // f for r in cfg["attn_res"]], latent_dim=[3, size, size])
    torch.manual_seed(1)
    m = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to(DEV).eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn((batch, 3, size, size), generator=g).to(DEV)
    t = torch.rand((batch,), generator=g).to(DEV)
    out = _both(m, x, t, monkeypatch)
    assert torch.isfinite(out[True]).all() and torch.equal(out[True], out[False])
    if batch == 16:
        samples = {}
        for fuse in (True, False):
            monkeypatch.setattr(E, "FUSE_EVAL_NORM", fuse)
            with torch.no_grad():
                samples[fuse] = m.generate(batch, 3, initial_noise=x)
        assert torch.equal(samples[True], samples[False])
