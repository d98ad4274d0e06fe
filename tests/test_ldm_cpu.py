"""The latent-diffusion restatement (oracle/ldm.py) against the fixtures recorded from the reference's own
`LatentDiffusion` code (tests/golden/{l0,l1}.safetensors, made by oracle/make_golden_ldm.py), and the host side of
vision_amd's LatentDiffusion that needs no GPU."""
import pytest
import torch

from conftest import load_golden, sub
from oracle import ldm as O
from oracle import restatement as R

CASES = ["l0", "l1"]


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return (request.param,) + load_golden(request.param)


def clone(d):
    return {k: v.clone() for k, v in d.items()}


def test_schedule_forward_and_taps(case):
    name, g, cfg, B = case
    n, s = R.diffusion_schedule(g["t"])
    torch.testing.assert_close(n, g["sched_n"], rtol=0, atol=1e-7)
    torch.testing.assert_close(s, g["sched_s"], rtol=0, atol=1e-7)
    out = O.unet_forward(clone(sub(g, "state0")), cfg, g["x_t"], g["sched_n"] ** 2, False)
    torch.testing.assert_close(out, g["fwd_eval"], rtol=1e-5, atol=1e-5)
    taps = {}
    sd = clone(sub(g, "state0"))
    loss = O.p_loss(sd, cfg, g["x0"], g["eps"], g["t"], True, None, taps)
    torch.testing.assert_close(loss.reshape(1), g["loss"], rtol=1e-5, atol=1e-6)
    for k, v in sub(g, "tap").items():
        torch.testing.assert_close(taps[k], v, rtol=1e-4, atol=1e-4)
    for k, v in sub(g, "bn1").items():
        torch.testing.assert_close(sd[k], v, rtol=1e-5, atol=1e-6)


def test_gradients_and_three_adamw_steps(case):
    name, g, cfg, B = case
    loss, grads = O.loss_and_grads(clone(sub(g, "state0")), cfg, g["x0"], g["eps"], g["t"])
    want = sub(g, "grad")
    assert set(grads) == set(want)
    big = max(float(v.abs().max()) for v in want.values())
    for k in want:
        assert float((grads[k] - want[k]).abs().max()) <= 2e-4 * float(want[k].abs().max()) + 1e-6 * big, k
    sd = clone(sub(g, "state0"))
    params = {k: sd[k] for k in want}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    losses = []
    for step in range(3):
        from oracle.make_golden import _draw
        eps, t = _draw(100 + step, g["x0"])
        loss, grads = O.loss_and_grads(sd, cfg, g["x0"], eps, t)
        # the statistics update of the training-mode forward is part of the state
        O.p_loss(sd, cfg, g["x0"], eps, t, True)
        R.adamw_step(params, grads, m, v2, step + 1, cfg["learning_rate"], cfg["weight_decay"])
        losses.append(float(loss))
    torch.testing.assert_close(torch.tensor(losses), g["loss_steps"], rtol=2e-3, atol=1e-4)


def test_validation_loss_and_sampler(case):
    name, g, cfg, B = case
    sd = clone(sub(g, "state3"))
    torch.manual_seed(200)
    eps = torch.randn_like(g["x0"])
    t = torch.rand((B, 1, 1, 1))
    torch.testing.assert_close(O.p_loss(sd, cfg, g["x0"], eps, t, False).reshape(1), g["val_loss"], rtol=1e-5, atol=1e-6)
    out = O.reverse_diffusion(sd, cfg, g["gen_noise"], 4)
    # the last step divides by a signal rate of 0.02: judged against the largest value, as for the DDPM sampler
    assert float((out - g["gen_T4"]).abs().max()) <= 1e-4 * float(g["gen_T4"].abs().max())


# ------------------------------------------------------------------ host side of the product module
def test_module_tree_has_the_reference_keys_and_shapes(case):
    name, g, cfg, B = case
    import vision_amd
    model = vision_amd.get_model("ldm", cfg)
    want = sub(g, "state0")
    got = model.state_dict()
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape and got[k].dtype == want[k].dtype, k
    torch.manual_seed(cfg["manual_seed"])
    seeded = vision_amd.get_model("ldm", cfg).state_dict()
    assert all(torch.equal(seeded[k], want[k]) for k in want)        # same initialisers in the same RNG order
    assert model.attn_resolutions == cfg["attn_res"] and model.patch_size == cfg["in_res"]


def test_configuration_errors():
    from vision_amd.plan import make_ldm_plan
    cfg = dict(h_dims=[32, 32, 64], down_index=[0, 1, 0], attn_res=[8], in_res=8, with_conv=True, dropout=0.0,
               diffusion_steps=4, sampling_period=1, grid=2)
    plan = make_ldm_plan(cfg)
    assert [l.kind for l in plan.layers] == ["res", "attn", "down", "res", "res", "attn", "res", "res", "up", "res", "attn"]
    for k in ("h_dims", "in_res", "grid"):
        bad = dict(cfg)
        del bad[k]
        with pytest.raises(KeyError):
            make_ldm_plan(bad)
