"""The VQ-VAE-UNet restatement (oracle/vqunet.py) against the fixture recorded from the unmodified reference
(tests/golden/vqunet.safetensors, made by oracle/make_golden_vqunet.py), and the host side of
vision_amd.VQVAEUNETModel that needs no GPU: module tree, state_dict keys, configuration errors."""
import json
import os

import pytest
import torch
from safetensors import safe_open

from oracle import vqunet as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vqunet.safetensors")


def load_case(name):
    out = {}
    with safe_open(GOLDEN, "pt") as f:
        cfg = json.loads(f.metadata()["configs"])[name]
        for k in f.keys():
            if k.startswith(name + "/"):
                out[k[len(name) + 1:]] = f.get_tensor(k)
    return cfg, out


def sub(d, prefix):
    return {k[len(prefix):]: v.clone() for k, v in d.items() if k.startswith(prefix)}


CASES = ["v0", "v1"]


@pytest.mark.parametrize("name", CASES)
def test_training_forward_matches_reference(name):
    cfg, g = load_case(name)
    sd = sub(g, "state/")
    taps = {}
    total, rec, vq_loss, x_rec, idx = O.loss(sd, cfg, g["x"], True, taps)
    assert torch.equal(idx, g["train/indices"])
    for k, v in sub(g, "train/tap/").items():
        if k in taps:                                      # the head's Sequential output is x_rec itself
            torch.testing.assert_close(taps[k], v, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x_rec, g["train/x_rec"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(vq_loss.reshape(1), g["train/vq_loss"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(total.reshape(1), g["train/loss"], rtol=1e-5, atol=1e-7)
    # train-mode BatchNorm moved the running statistics exactly as the reference's did (the optimiser leaves buffers alone)
    after = sub(g, "after/")
    for k in sd:
        if not O.is_param(k):
            torch.testing.assert_close(sd[k], after[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_gradients_and_adam_step_match_reference(name):
    cfg, g = load_case(name)
    sd = sub(g, "state/")
    total, rec, vq_loss, x_rec, idx, grads = O.loss_and_grads(sd, cfg, g["x"])
    want = sub(g, "grad/")
    assert set(grads) == set(want)
    for k in want:
        scale = float(want[k].abs().max()) + 1e-12
        assert float((grads[k] - want[k]).abs().max()) <= 2e-4 * scale + 1e-7, k
    params = {k: sd[k] for k in want}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    O.adam_l2_step(params, want, m, v2, 1, cfg["learning_rate"], cfg["weight_decay"])
    after = sub(g, "after/")
    for k in params:
        torch.testing.assert_close(params[k], after[k], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", CASES)
def test_eval_forward_matches_reference(name):
    cfg, g = load_case(name)
    sd = sub(g, "after/")
    taps = {}
    total, rec, vq_loss, x_rec, idx = O.loss(sd, cfg, g["x2"], False, taps)
    assert torch.equal(idx, g["eval/indices"])
    torch.testing.assert_close(taps["latents"], g["eval/latents"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x_rec, g["eval/x_rec"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(total.reshape(1), g["eval/loss"], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ host side of the product module
def test_module_tree_has_the_reference_keys_and_shapes():
    import vision_amd
    for name in CASES:
        cfg, g = load_case(name)
        model = vision_amd.get_model("vqvaeunet", cfg)
        want = sub(g, "state/")
        got = model.state_dict()
        assert sorted(got) == sorted(want)                # safetensors stores keys sorted; creation order is checked against the reference when the fixture is made
        for k in want:
            assert got[k].shape == want[k].shape and got[k].dtype == want[k].dtype, k
        model.load_state_dict(want)                      # strict


def test_butterfly_configuration_builds_without_attention():
    from vision_amd.plan import make_vq_plan
    cfg = dict(h_dims=[3, 64, 64, 128, 128, 256, 256], down_index=[0, 1, 0, 1, 1, 1, 1], patch_size=224,
               sampling_period=1, embed_dim=256, num_embed=2048, attn_res=[7])
    plan = make_vq_plan(cfg)               # reference configs/butterfly_vqvaeunet.yaml:7-13: 7 is reached after the last check
    kinds = [l.kind for l in plan.section("Downsampling")]
    assert kinds == ["res", "down", "res", "res", "down", "res", "down", "res", "down", "res", "down"]
    assert [l.kind for l in plan.section("Upsampling")] == ["up", "res", "up", "res", "up", "res", "up", "res", "res", "up", "res"]
    assert plan.section("Downsampling")[-1].res_out == 7 and plan.end_channels == 3 and plan.time_features == 0


def test_configuration_errors():
    from vision_amd.plan import make_vq_plan
    cfg = dict(h_dims=[3, 32, 64], down_index=[0, 1, 1], patch_size=16, sampling_period=1, embed_dim=64, num_embed=48,
               attn_res=[])
    for k in ("h_dims", "num_embed", "attn_res"):
        bad = dict(cfg)
        del bad[k]
        with pytest.raises(KeyError):
            make_vq_plan(bad)
    # an AttnBlock where attn_res lists the resolution: before the level's DownBlock, after the level's UpBlock
    plan = make_vq_plan(dict(cfg, attn_res=[8]))
    assert [l.kind for l in plan.section("Downsampling")] == ["res", "down", "res", "attn", "down"]
    assert [l.kind for l in plan.section("Upsampling")] == ["up", "res", "attn", "up", "res"]
    with pytest.raises(ValueError):
        make_vq_plan(dict(cfg, embed_dim=32))
