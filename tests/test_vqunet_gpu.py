"""Parity of the HIP VQVAEUNETModel against vectors recorded from the unmodified reference
(tests/golden/vqunet.safetensors): training-mode forward (codes, reconstruction, both loss terms), every parameter
gradient, BatchNorm running statistics, one Adam step with L2 weight decay, an eval-mode forward of the updated model,
and the reference's own two-call surface (forward + loss_function, encode / quantize / decode).

Tolerances.  fp32 compute: codes identical (the fixture's codes are apart by > 1e-4 of their distance); reconstruction
1e-5 absolute (values in (0, 1)); losses 1e-5 relative; gradients 5e-4 relative Frobenius norm per tensor and over all
parameters together; parameters after the step: the step of 1e-3 per element (lr) may flip where a gradient is within
rounding of zero, so they are judged by the share of elements that moved the reference's way (>= 99.5 %).
bf16 compute (fp32 master weights, fp32 codebook lookup): at most 5 % of the codes differ, reconstruction 2e-2
absolute, losses 2e-2 relative, gradients 1e-1 over all parameters together.
Parameters whose exact gradient is zero (conv1 biases feeding a train-mode BatchNorm) must only stay negligible.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import vision_amd                         # noqa: E402
from vision_amd._lib import call          # noqa: E402
from oracle import vqunet as O            # noqa: E402  (checker only)
from test_vqunet_cpu import CASES, load_case, sub     # noqa: E402

DEV = torch.device("cuda:0")
DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {
    torch.float32: dict(codes=0.0, rec=1e-5, loss=1e-5, grad=5e-4, grad_all=5e-4, bn=1e-5),
    torch.bfloat16: dict(codes=0.05, rec=2e-2, loss=2e-2, grad=None, grad_all=1e-1, bn=1e-2),
}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def make_model(cfg, state, dtype):
    m = vision_amd.get_model("vqvaeunet", cfg, compute_dtype=dtype)
    m.load_state_dict(state)
    return m.to(DEV)


@pytest.fixture(scope="module", params=list(DTYPES))
def dtype(request):
    return DTYPES[request.param]


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return (request.param,) + load_case(request.param)


def zero_grad_keys(grads):
    return {k for k in grads if k.endswith(".block1.0.2.bias")}


def test_training_step(case, dtype):
    name, cfg, g = case
    tol = TOL[dtype]
    m = make_model(cfg, sub(g, "state/"), dtype).train()
    x = g["x"].to(DEV)
    opt, _ = m.configure_optimizers()
    opt = opt[0]
    loss = m.training_step((x, None), 0)
    eng = m._ensure_engine()
    wrong = (eng.last_indices.cpu() != g["train/indices"]).float().mean().item()
    assert wrong <= tol["codes"], f"{wrong:.3f} of the codes differ"
    assert abs(loss.item() - g["train/loss"].item()) <= tol["loss"] * g["train/loss"].item()
    loss.backward()
    want = sub(g, "grad/")
    zero = zero_grad_keys(want)
    got = {k: p.grad for k, p in m.named_parameters()}
    assert set(got) == set(want)
    num = den = 0.0
    for k in want:
        if k in zero:
            assert got[k].abs().max().item() <= 1e-6 + 1e-3 * max(v.abs().max().item() for v in want.values()), k
            continue
        if tol["grad"] is not None:
            assert rel(got[k], want[k]) <= tol["grad"], (k, rel(got[k], want[k]))
        num += (got[k].double().cpu() - want[k].double()).pow(2).sum().item()
        den += want[k].double().pow(2).sum().item()
    assert (num / den) ** 0.5 <= tol["grad_all"], (num / den) ** 0.5
    # running statistics
    after = sub(g, "after/")
    sd = m.state_dict()
    for k, v in after.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert (sd[k].cpu() - v).abs().max().item() <= tol["bn"] * (1.0 + v.abs().max().item()), k
        elif k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
    # the optimiser step: from the reference's gradients, so that only the update rule is judged here
    for k, p in m.named_parameters():
        p.grad.copy_(want[k].to(DEV))
    opt.step()
    moved = total = 0
    before = sub(g, "state/")
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.detach().cpu(), after[k], rtol=1e-5, atol=2e-6)
        total += p.numel()
        moved += ((p.detach().cpu() - before[k]).sign() == (after[k] - before[k]).sign()).sum().item()
    assert moved / total >= 0.995


def test_eval_forward(case, dtype):
    name, cfg, g = case
    tol = TOL[dtype]
    m = make_model(cfg, sub(g, "after/"), dtype).eval()
    x2 = g["x2"].to(DEV)
    with torch.no_grad():
        x_rec, vq_loss = m(x2)
        loss = m.loss_function(x2, x_rec, vq_loss)
    eng = m._ensure_engine()
    wrong = (eng.last_indices.cpu() != g["eval/indices"]).float().mean().item()
    assert wrong <= tol["codes"], f"{wrong:.3f} of the codes differ"
    assert x_rec.shape == g["eval/x_rec"].shape and x_rec.dtype == torch.float32
    assert (x_rec.cpu() - g["eval/x_rec"]).abs().max().item() <= tol["rec"] * (1 if dtype == torch.float32 else 3)
    assert abs(loss.item() - g["eval/loss"].item()) <= tol["loss"] * g["eval/loss"].item()
    # the fused objective of validation_step gives the same number as forward + loss_function
    with torch.no_grad():
        fused = m._step_loss((x2, None))
    assert abs(fused.item() - loss.item()) <= 1e-6 * abs(loss.item()) + 1e-7


def test_two_call_surface_gives_the_same_gradients(case):
    """forward(x) + loss_function(...) (the reference's training_step, models/vqvaeunet.py:241-247, with torch's
    mse_loss on the returned tensors) against the fused objective: same loss, same gradients."""
    name, cfg, g = case
    x = g["x"].to(DEV)
    grads = []
    for fused in (True, False):
        m = make_model(cfg, sub(g, "state/"), torch.float32).train()
        if fused:
            loss = m._step_loss((x, None))
        else:
            x_rec, vq_loss = m(x)
            loss = m.loss_function(x, x_rec, vq_loss)
        loss.backward()
        grads.append((loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * grads[0][0]
    for k in grads[0][1]:
        a, b = grads[0][1][k], grads[1][1][k]
        assert (a - b).abs().max().item() <= 1e-5 * (b.abs().max().item() + 1e-8) + 1e-9, k


def test_encode_quantize_decode_compose_to_forward(case, dtype):
    name, cfg, g = case
    m = make_model(cfg, sub(g, "after/"), dtype).eval()
    x2 = g["x2"].to(DEV)
    with torch.no_grad():
        x_rec, vq_loss = m(x2)
        h, skip = m.encode(x2)
        n_skip = len(skip)
        q, vq2, codes = m.quantize(h)
        again = m.decode(q, skip)
    assert len(skip) == n_skip == len(m.Downsampling)               # decode() leaves the caller's list alone
    assert h.shape == g["eval/latents"].shape
    assert codes.shape == (x2.shape[0],) + tuple(h.shape[2:])
    if dtype == torch.float32:
        assert (h.cpu() - g["eval/latents"]).abs().max().item() <= 1e-4
        assert torch.equal(codes.view(-1).cpu(), g["eval/indices"])
        assert abs(vq2.item() - vq_loss.item()) <= 1e-6 * vq_loss.item()
        assert (again - x_rec).abs().max().item() <= 1e-6
    else:
        # the latents crossed the API boundary in the compute dtype, the fused forward keeps them in registers / fp32
        assert (again - x_rec).abs().max().item() <= 3e-2


def test_refuses_what_it_cannot_do(case):
    name, cfg, g = case
    m = make_model(cfg, sub(g, "state/"), torch.float32).train()
    x = g["x"].to(DEV).requires_grad_(True)
    with pytest.raises(RuntimeError, match="parameter gradients only"):
        m(x)
    with pytest.raises(ValueError, match="encoder outputs"):
        m.decode(torch.zeros(1, cfg["embed_dim"], 4, 4, device=DEV), [])
    # an AttnBlock at a level whose block changes the channel count: the decoder's block output (32 channels) meets the
    # encoder's (64) in `h += skip[index]`; the reference fails on the shapes there, this engine before the launch
    bad = dict(cfg, h_dims=[3, 32, 64], down_index=[0, 1, 1], attn_res=[8], embed_dim=64)
    mb = vision_amd.get_model("vqvaeunet", bad, compute_dtype=torch.float32).to(DEV).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="do not mirror"):
        mb(g["x"].to(DEV))


# ------------------------------------------------------------------ the two kernels this model adds
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_sigmoid_mse_kernels(dt):
    from vision_amd import _lib
    dtype = DTYPES[dt]
    code = _lib.F32 if dtype == torch.float32 else _lib.BF16
    gen = torch.Generator(device="cpu").manual_seed(3)
    B, C, H, W = 3, 3, 37, 21
    h = (torch.randn((B, H, W, C), generator=gen) * 3).to(dtype).to(DEV)
    target = torch.rand((B, C, H, W), generator=gen).to(DEV)
    xrec = torch.empty((B, C, H, W), device=DEV)
    part = torch.empty(_lib.lib.vamd_l1_chunks(B * H * W), device=DEV)
    loss = torch.empty((), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    call("vamd_sigmoid_mse_fwd", h.data_ptr(), target.data_ptr(), xrec.data_ptr(), part.data_ptr(), loss.data_ptr(), B, C, H, W, code, st)
    href = h.float().permute(0, 3, 1, 2).double().requires_grad_(True)
    yref = torch.sigmoid(href)
    lref = torch.nn.functional.mse_loss(yref, target.double())
    assert (xrec.double() - yref).abs().max().item() <= 2e-7
    assert abs(loss.item() - lref.item()) <= 2e-6 * lref.item()
    only = torch.empty_like(xrec)
    call("vamd_sigmoid_mse_fwd", h.data_ptr(), None, only.data_ptr(), None, None, B, C, H, W, code, st)
    assert torch.equal(only, xrec)
    g_loss = torch.tensor([1.7], device=DEV)
    g_x = torch.randn((B, C, H, W), generator=gen).to(DEV)
    (1.7 * lref + (yref * g_x.double()).sum()).backward()
    dh = torch.empty_like(h)
    call("vamd_sigmoid_mse_bwd", xrec.data_ptr(), target.data_ptr(), g_loss.data_ptr(), g_x.data_ptr(), dh.data_ptr(), B, C, H, W, code, st)
    want = href.grad.permute(0, 2, 3, 1)
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert (dh.double() - want).abs().max().item() <= tol * want.abs().max().item()


def test_adam_with_l2_decay_matches_torch():
    n = 10007
    gen = torch.Generator().manual_seed(4)
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=1e-2)
    pad = (n + 3) // 4 * 4
    p = torch.zeros(pad, device=DEV)
    p[:n] = p0.to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    st = torch.cuda.current_stream().cuda_stream
    for step in range(1, 4):
        gr = torch.randn(n, generator=gen)
        ref.grad = gr.clone()
        opt.step()
        gbuf = torch.zeros(pad, device=DEV)
        gbuf[:n] = gr.to(DEV)
        call("vamd_adam_l2", p.data_ptr(), gbuf.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, None, 0.9, 0.999, 1e-8, 1e-2,
             step, 1.0, st)
        torch.testing.assert_close(p[:n].cpu(), ref.detach(), rtol=2e-6, atol=2e-7)


# ------------------------------------------------------------------ the shipped configuration's sizes
def test_butterfly_configuration_trains():
    """reference configs/butterfly_vqvaeunet.yaml at its own sizes (224 x 224, batch 16, 2048 codes of 256): two
    optimiser steps in both compute modes against two steps of the oracle on the host (about 4 s each).  The loss RISES
    from the first step to the second in the reference (0.179 -> 0.818 with these seeds: the first Adam step moves the
    latents away from a codebook that starts within 1/2048 of zero), so the second loss checks the whole update."""
    cfg = dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, h_dims=[3, 64, 64, 128, 128, 256, 256],
               down_index=[0, 1, 0, 1, 1, 1, 1], patch_size=224, sampling_period=1, embed_dim=256, num_embed=2048,
               attn_res=[7], weight_decay=1e-3, scheduler_gamma=0.99)
    torch.manual_seed(42)
    state = {k: v.clone() for k, v in vision_amd.get_model("vqvaeunet", cfg).state_dict().items()}
    x = torch.rand((16, 3, 224, 224), generator=torch.Generator().manual_seed(0))

    sd = {k: v.clone() for k, v in state.items()}
    params = {k: sd[k] for k in sd if O.is_param(k)}
    m1 = {k: torch.zeros_like(v) for k, v in params.items()}
    m2 = {k: torch.zeros_like(v) for k, v in params.items()}
    want = []
    for step in (1, 2):
        total, _, _, _, _, grads = O.loss_and_grads(sd, cfg, x)
        O.adam_l2_step(params, grads, m1, m2, step, cfg["learning_rate"], cfg["weight_decay"])
        want.append(total.item())
    assert want[1] > want[0]

    xd = x.to(DEV)
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 2e-2)):
        m = make_model(cfg, state, dtype).train()
        opt = m.configure_optimizers()[0][0]
        got = []
        for _ in range(2):
            loss = m.training_step((xd, None), 0)
            loss.backward()
            opt.step()
            got.append(loss.item())
        for a, b in zip(got, want):
            assert abs(a - b) <= tol * b, (dtype, got, want)
        assert m._ensure_engine().last_indices.shape == (16 * 7 * 7,)
        m.eval()
        with torch.no_grad():
            x_rec, vq_loss = m(xd)
        assert x_rec.shape == xd.shape and 0.0 <= x_rec.min().item() and x_rec.max().item() <= 1.0 and vq_loss.item() >= 0.0


def test_run_flow_of_the_first_stage(tmp_path, capsys):
    """run.py end to end with `name: vqvaeunet` on a small config and synthetic images: metrics, a reconstruction grid
    after every epoch (reference models/vqvaeunet.py:278-306), top-k + last checkpoints with the reference's keys, and a
    second invocation that resumes from version_0 (reference run.py:48-59)."""
    import yaml
    import run
    from safetensors.torch import load_file
    cfg = load_case("v0")[0]

    def config(max_epochs):
        return {
            "model_config": cfg,
            "data_config": {"patch_size": cfg["patch_size"], "data_dir": "unused", "batch_size": 4, "num_workers": 0,
                            "use_manual_split": True, "length": 60},
            "trainer_config": {"accelerator": "gpu", "devices": 1, "max_epochs": max_epochs, "precision": "16-mixed"},
            "log_config": {"name": "v0_vqvaeunet", "save_dir": str(tmp_path / "logs"),
                           "check_dir": str(tmp_path / "logs" / "check"), "checkpoint": True, "model_summary": False},
        }

    path = tmp_path / "v0.yaml"
    path.write_text(yaml.safe_dump(config(2), sort_keys=False))
    model, logger = run.main(["--config", str(path), "--synthetic"])
    out = capsys.readouterr().out
    assert isinstance(model, vision_amd.VQVAEUNETModel) and "No checkpoint found" in out and "TeL" in out
    v0 = tmp_path / "logs" / "v0_vqvaeunet" / "version_0"
    rows = [json.loads(l) for l in open(v0 / "metrics.jsonl")]
    assert len(rows) == 2 and all({"TL", "VL", "lr", "step"} <= set(r) for r in rows)
    assert all(0.0 < r["TL"] < 10.0 and 0.0 < r["VL"] < 10.0 for r in rows)
    for e in (0, 1):
        assert (v0 / "Reconstructions" / f"recons_v0_vqvaeunet_Epoch_{e}.png").exists()
    last = load_file(str(v0 / "checkpoints" / "last.safetensors"))
    final = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert set(last) == set(final) and all(torch.equal(last[k], final[k]) for k in last)
    assert "vq_layer.embedding.weight" in last and "Downsampling.0.layers.0.linear.weight" in last
    path.write_text(yaml.safe_dump(config(0), sort_keys=False))
    model2, logger2 = run.main(["--config", str(path), "--synthetic"])
    out = capsys.readouterr().out
    assert logger2.version == 1 and "Loaded checkpoint from" in out
    assert all(torch.equal(last[k], v.detach().cpu()) for k, v in model2.state_dict().items())


def test_eval_mode_gradients(case):
    """model.eval() followed by loss.backward(): BatchNorm uses its running statistics, so its input gradient has no mean /
    variance terms and the biases in front of it (zero gradient in training mode) are live; against the oracle in eval mode."""
    name, cfg, g = case
    sd = sub(g, "after/")
    m = make_model(cfg, sd, torch.float32).eval()
    loss = m._step_loss((g["x2"].to(DEV), None))
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if O.is_param(k)}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaves)
    total, _, _, _, idx = O.loss(work, cfg, g["x2"], False)
    want = dict(zip(leaves, torch.autograd.grad(total, list(leaves.values()), allow_unused=True)))
    assert torch.equal(m._ensure_engine().last_indices.cpu(), idx)
    assert abs(loss.item() - total.item()) <= 1e-5 * total.item()
    got = {k: p.grad for k, p in m.named_parameters()}
    for k, w in want.items():
        w = torch.zeros_like(leaves[k]) if w is None else w
        assert rel(got[k], w) <= 5e-4, (k, rel(got[k], w))
    k1 = next(k for k in want if k.endswith(".block1.0.2.bias"))
    assert want[k1].abs().max().item() > 0
    # the running statistics did not move
    for k, v in sd.items():
        if "running_" in k:
            assert torch.equal(m.state_dict()[k].cpu(), v), k


def test_add_cast_kernel():
    from vision_amd import _lib
    code = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
    gen = torch.Generator().manual_seed(8)
    n = 12345
    st = torch.cuda.current_stream().cuda_stream
    for ta in code:
        for tb in (None,) + tuple(code):
            for to in code:
                a = torch.randn(n, generator=gen).to(ta).to(DEV)
                b = None if tb is None else torch.randn(n, generator=gen).to(tb).to(DEV)
                out = torch.empty(n, dtype=to, device=DEV)
                call("vamd_add_cast", a.data_ptr(), code[ta], None if b is None else b.data_ptr(), 0 if tb is None else code[tb],
                     out.data_ptr(), code[to], n, st)
                want = (a.float() + (b.float() if b is not None else 0.0)).to(to)
                assert torch.equal(out, want), (ta, tb, to)
