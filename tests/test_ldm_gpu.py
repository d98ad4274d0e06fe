"""Parity of the HIP LatentDiffusion against vectors recorded from the reference's own models/ldm.py code
(tests/golden/{l0,l1}.safetensors): eval forward, training objective, every parameter gradient, BatchNorm statistics,
three AdamW steps, validation loss, fixed-noise sampling; and latent diffusion over the VQ-VAE-UNet first stage end to
end (images -> latents -> objective; sampled latents -> codebook lookup -> decoder).

How gradients are judged (as for the DDPM model, tests/test_model_gpu.py): the L1 objective has a discontinuous
gradient and the time embedding turns 1 ulp of n^2 into ~5e-4, so the strict per-tensor check feeds the reference's
x_t and n^2 bits through forward(x, t), uses a smooth objective sum(out * w) with a fixed random w, and compares with
the same function evaluated in fp64 by the oracle:
  fp32 compute: per tensor <= max(2e-5, 4 x the error of the oracle's own fp32 evaluation); all parameters together
                <= max(1e-5, 4 x that figure)
  bf16 compute: all parameters together <= 8e-2.
The whole objective p_loss: against the recorded fp32 gradients 2e-2 (fp32) / 1e-1 (bf16) over all parameters, and in fp32
1e-5 against the oracle evaluated on the device's own schedule values (see test_loss_grads_bnstats).  Other tolerances: forward 1e-4 / 3e-2 relative, loss 1e-5 / 2e-2, sampler 1e-3 / 8e-2 of max|ref|.
Parameters whose exact gradient is zero (`*.block1.0.2.bias` and `*.linear_temb.linear.bias`: both feed a train-mode
BatchNorm, whose mean subtraction cancels them) must only stay negligible.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import vision_amd                         # noqa: E402
from conftest import load_golden, sub     # noqa: E402
from oracle import ldm as O               # noqa: E402  (checker only)

DEV = torch.device("cuda:0")
DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {
    torch.float32: dict(fwd=1e-4, loss=1e-5, grad_all=3e-3, bn=1e-4, gen=1e-3),
    torch.bfloat16: dict(fwd=3e-2, loss=2e-2, grad_all=1e-1, bn=1e-2, gen=8e-2),
}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def cat(d, keys):
    return torch.cat([d[k].detach().double().flatten().cpu() for k in keys])


def zero_keys(keys):
    return {k for k in keys if k.endswith((".block1.0.2.bias", ".linear_temb.linear.bias"))}


def make_model(cfg, state, dtype):
    m = vision_amd.get_model("ldm", cfg, compute_dtype=dtype)
    m.load_state_dict(state)
    return m.to(DEV)


@pytest.fixture(scope="module", params=list(DTYPES))
def dtype(request):
    return DTYPES[request.param]


@pytest.fixture(scope="module", params=["l0", "l1"])
def case(request):
    return (request.param,) + load_golden(request.param)


def test_forward_eval(case, dtype):
    name, g, cfg, B = case
    m = make_model(cfg, sub(g, "state0"), dtype).eval()
    with torch.no_grad():
        out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))
    assert out.shape == g["fwd_eval"].shape and out.dtype == torch.float32
    assert rel(out, g["fwd_eval"]) < TOL[dtype]["fwd"]


def test_backward_exact_inputs(case, dtype):
    name, g, cfg, B = case
    sd = sub(g, "state0")
    w = torch.randn(g["x_t"].shape, generator=torch.Generator().manual_seed(5))
    n2 = g["sched_n"] ** 2

    def truth(dt):
        leaves = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items() if O.is_param(k)}
        work = {k: (v.to(dt).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        work.update(leaves)
        out = O.unet_forward(work, cfg, g["x_t"].to(dt), n2.to(dt), True)
        gs = torch.autograd.grad((out * w.to(dt)).sum(), list(leaves.values()), allow_unused=True)
        return out.detach(), {k: (v if v is not None else torch.zeros_like(leaves[k])) for k, v in zip(leaves, gs)}

    out64, g64 = truth(torch.float64)
    _, g32 = truth(torch.float32)
    m = make_model(cfg, sd, dtype).train()
    out = m(g["x_t"].to(DEV), n2.to(DEV))
    assert rel(out, out64) < (1e-5 if dtype == torch.float32 else 3e-2)
    (out * w.to(DEV)).sum().backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    zero = zero_keys(g64)
    live = sorted(k for k in g64 if k not in zero)
    scale = max(v.abs().max().item() for v in g64.values())
    failures = []
    for k in g64:
        assert torch.isfinite(grads[k]).all(), k
        if k in zero:
            if grads[k].abs().max().item() > (1e-4 if dtype == torch.float32 else 5e-2) * scale:
                failures.append((k, "nonzero", grads[k].abs().max().item()))
        elif dtype == torch.float32:
            e, bound = rel(grads[k], g64[k]), max(2e-5, 4 * rel(g32[k], g64[k]))
            if e > bound:
                failures.append((k, e, bound))
    assert not failures, failures
    e_all = rel(cat(grads, live), cat(g64, live))
    bound = max(1e-5, 4 * rel(cat(g32, live), cat(g64, live))) if dtype == torch.float32 else 8e-2
    assert e_all < bound, (e_all, bound)


def test_loss_grads_bnstats(case, dtype):
    """The whole objective p_loss (schedule and noising on the device).  The device's sin / cos are within 2 ulp of
    torch's; the function itself turns that into up to 8e-3 of gradient error (one LeakyReLU gate of the time-embedding
    MLP flips: each of its few hundred elements reaches every block), so against the RECORDED gradients the bound is loose,
    and the tight fp32 check evaluates the oracle on the device's own schedule values (measured 4e-7 ... 6e-7;
    tests/diag_ldm_flips_gpu.py)."""
    name, g, cfg, B = case
    tol = TOL[dtype]
    m = make_model(cfg, sub(g, "state0"), dtype).train()
    loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < tol["loss"] * abs(g["loss"].item())
    grads = {k: p.grad for k, p in m.named_parameters()}
    ref = sub(g, "grad")
    live = sorted(k for k in ref if k not in zero_keys(ref))
    e_all = rel(cat(grads, live), cat(ref, live))
    assert e_all < (2e-2 if dtype == torch.float32 else tol["grad_all"]), e_all
    if dtype == torch.float32:
        n_dev, s_dev = (v.cpu() for v in m.diffusion_schedule(g["t"].to(DEV)))
        assert (n_dev - g["sched_n"]).abs().max() < 3e-7 and (s_dev - g["sched_s"]).abs().max() < 3e-7
        sd0 = sub(g, "state0")
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if O.is_param(k)}
        work = {k: v.clone() for k, v in sd0.items()}
        work.update(leaves)
        pred = O.unet_forward(work, cfg, s_dev * g["x0"] + n_dev * g["eps"], n_dev ** 2, True)
        own = dict(zip(leaves, torch.autograd.grad(torch.nn.functional.l1_loss(pred, g["eps"]), list(leaves.values()))))
        assert rel(cat(grads, live), cat(own, live)) < 1e-5
    sd = m.state_dict()
    for k, v in sub(g, "bn1").items():
        if v.dtype == torch.int64:
            assert int(sd[k]) == int(v), k
        else:
            assert rel(sd[k], v) < tol["bn"], k


def test_three_adamw_steps_then_eval_and_sample(case, dtype):
    name, g, cfg, B = case
    tol = TOL[dtype]
    m = make_model(cfg, sub(g, "state0"), dtype).train()
    opt = m.configure_optimizers()[0][0]
    x0 = g["x0"].to(DEV)
    losses = []
    for step in range(3):
        torch.manual_seed(100 + step)                       # CPU generator: the reference's draws
        eps = torch.randn_like(g["x0"])
        t = torch.rand((B, 1, 1, 1))
        loss = m.p_loss(x0, eps.to(DEV), t.to(DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    for a, b in zip(losses, g["loss_steps"].tolist()):
        assert abs(a - b) < (2e-3 if dtype == torch.float32 else 3e-2) * abs(b), (losses, g["loss_steps"].tolist())
    m2 = make_model(cfg, sub(g, "state3"), dtype).eval()
    with torch.no_grad():
        torch.manual_seed(200)
        eps = torch.randn_like(g["x0"])
        t = torch.rand((B, 1, 1, 1))
        val = m2.p_loss(x0, eps.to(DEV), t.to(DEV))
        assert abs(val.item() - g["val_loss"].item()) < tol["loss"] * abs(g["val_loss"].item())
        out = m2.generate(2, 4, initial_noise=g["gen_noise"].to(DEV))
        ref = g["gen_T4"]
        assert (out.cpu() - ref).abs().max().item() < tol["gen"] * ref.abs().max().item()
        assert m2.generate(3, 2).shape == (3, cfg["h_dims"][0], cfg["in_res"], cfg["in_res"])     # own noise: latent-shaped


def test_latent_diffusion_over_the_first_stage():
    """configs/butterfly_vqvaeunet.yaml first stage at its own sizes (224 x 224 -> 256 x 7 x 7 latents, 2048 codes) with
    a latent UNet on top: the step methods take IMAGES; their loss equals p_loss on the first stage's encoder output;
    sampled latents go through the codebook lookup -- codes identical to an exact fp64 nearest-code search, up to
    distances within fp32 rounding of each other -- and the decoder (zero skips), giving images in (0, 1)."""
    fs_cfg = dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, h_dims=[3, 64, 64, 128, 128, 256, 256],
                  down_index=[0, 1, 0, 1, 1, 1, 1], patch_size=224, sampling_period=1, embed_dim=256, num_embed=2048,
                  attn_res=[7], weight_decay=1e-3, scheduler_gamma=0.99)
    ldm_cfg = dict(name="ldm", learning_rate=1e-4, manual_seed=42, h_dims=[256, 256, 256], down_index=[0, 0, 0],
                   attn_res=[7], in_res=7, with_conv=True, dropout=0.1, diffusion_steps=4, sampling_period=1, grid=2,
                   weight_decay=1e-5, scheduler_gamma=0.99)
    torch.manual_seed(42)
    fs = vision_amd.get_model("vqvaeunet", fs_cfg).to(DEV).eval()
    with torch.no_grad():
        fs.vq_layer.embedding.weight.normal_(0.0, 1.0, generator=None)        # a spread-out codebook, as after training
    torch.manual_seed(43)
    m = vision_amd.get_model("ldm", ldm_cfg).to(DEV).train()
    m.set_first_stage(fs)
    x = torch.rand((8, 3, 224, 224), generator=torch.Generator().manual_seed(0)).to(DEV)
    z = m.encode_images(x)
    assert z.shape == (8, 256, 7, 7) and not z.requires_grad
    opt = m.configure_optimizers()[0][0]
    before = {k: v.clone() for k, v in fs.state_dict().items()}
    torch.manual_seed(7)
    loss = m.training_step((x, None), 0)
    m.eval()
    with torch.no_grad():
        torch.manual_seed(7)
        a = m.validation_step((x, None), 0)
        va = m.logged["VL"].item()
        torch.manual_seed(7)
        vb = m.p_loss(z).item()
    assert abs(va - vb) <= 1e-6 * abs(vb)
    m.train()
    loss.backward()
    opt.step()
    assert 0.0 < loss.item() < 10.0
    assert all(torch.equal(before[k], v) for k, v in fs.state_dict().items())       # the first stage is not trained
    assert not any(k.startswith("_first_stage") or "vq_layer" in k for k in m.state_dict())
    m.eval()
    lat = m.generate(4, 4, initial_noise=torch.randn((4, 256, 7, 7), generator=torch.Generator().manual_seed(1)).to(DEV))
    img = m.decode_latents(lat)
    assert img.shape == (4, 3, 224, 224) and 0.0 <= img.min().item() and img.max().item() <= 1.0
    E = fs.vq_layer.embedding.weight.detach().double()
    zf = lat.permute(0, 2, 3, 1).reshape(-1, 256).double()
    d = (zf * zf).sum(1, keepdim=True) + (E * E).sum(1) - 2 * zf @ E.t()
    best = d.min(1).values
    chosen = d.gather(1, m.last_codes.view(-1, 1)).squeeze(1)
    assert m.last_codes.shape == (4, 7, 7)
    assert ((chosen - best) <= 1e-5 * best.abs().clamp_min(1.0)).all()
    assert (m.last_codes.view(-1) == d.argmin(1)).float().mean().item() >= 0.98


def test_first_stage_encoding_started_ahead_changes_nothing():
    """LatentDiffusion.encode_ahead(next batch) while the current step runs (what the trainer's loop and bench.py do)
    against steps that encode their own batch: identical losses and parameters after four optimiser steps; a batch that
    was announced is encoded once, one that was not is encoded by its step; latent batches are left alone."""
    from vision_amd.trainer import _with_lookahead
    fs_cfg = dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, h_dims=[3, 32, 64, 64], down_index=[0, 1, 1, 1],
                  patch_size=32, sampling_period=1, embed_dim=64, num_embed=128, attn_res=[4], weight_decay=1e-3,
                  scheduler_gamma=0.99)
    ldm_cfg = dict(name="ldm", learning_rate=1e-3, manual_seed=42, h_dims=[64, 64, 64], down_index=[0, 0, 0], attn_res=[4],
                   in_res=4, with_conv=True, dropout=0.1, diffusion_steps=4, sampling_period=1, grid=2, weight_decay=1e-5,
                   scheduler_gamma=0.99)
    gen = torch.Generator().manual_seed(2)
    batches = [(torch.rand((4, 3, 32, 32), generator=gen).to(DEV), None) for _ in range(4)]
    out = []
    for ahead in (True, False):
        torch.manual_seed(42)
        fs = vision_amd.get_model("vqvaeunet", fs_cfg).to(DEV).eval()
        torch.manual_seed(43)
        m = vision_amd.get_model("ldm", ldm_cfg).to(DEV).train()
        m.set_first_stage(fs)
        opt = m.configure_optimizers()[0][0]
        calls = []
        plain = m.encode_images
        m.encode_images = lambda x: (calls.append(id(x)), plain(x))[1]
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        losses = []
        for i, b in enumerate(_with_lookahead(batches, m.encode_ahead if ahead else None)):
            loss = m.training_step(b, i)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        assert calls == [id(b[0]) for b in ([batches[1], batches[0], batches[2], batches[3]] if ahead else batches)]
        assert not m._ahead
        out.append((torch.stack(losses).cpu(), m._ensure_engine().flat_p.clone()))
    (l0, p0), (l1, p1) = out
    assert torch.equal(l0, l1) and torch.equal(p0, p1)
    z = torch.randn((4, 64, 4, 4), device=DEV)
    m.encode_ahead((z, None))                                  # latents: nothing to encode
    assert not m._ahead


def test_run_flow_of_latent_diffusion(tmp_path, capsys):
    """run.py with `name: ldm` and a `first_stage_config`: image batches from the data module, latents from the (here
    untrained) first stage, metrics, decoded sample grids, a checkpoint that holds the latent UNet's keys only."""
    import json
    import yaml
    import run
    from safetensors.torch import load_file
    fs_cfg = dict(name="vqvaeunet", learning_rate=1e-3, manual_seed=42, h_dims=[3, 32, 32, 64], down_index=[0, 1, 0, 1],
                  patch_size=16, num_embed=64, embed_dim=64, sampling_period=1, attn_res=[], weight_decay=1e-3,
                  scheduler_gamma=0.99)
    fs_path = tmp_path / "fs.yaml"
    fs_path.write_text(yaml.safe_dump({"model_config": fs_cfg}, sort_keys=False))
    cfg = {
        "model_config": dict(name="ldm", learning_rate=1e-3, manual_seed=42, with_conv=True, dropout=0.0, scheduler_gamma=0.99,
                             h_dims=[64, 32, 32], down_index=[0, 0, 0], attn_res=[], in_res=4, sampling_period=1,
                             weight_decay=1e-2, diffusion_steps=4, grid=2, first_stage_config=str(fs_path),
                             first_stage_checkpoint=None),
        "data_config": {"patch_size": 16, "data_dir": "unused", "batch_size": 4, "num_workers": 0,
                        "use_manual_split": True, "length": 40},
        "trainer_config": {"accelerator": "gpu", "devices": 1, "max_epochs": 2, "precision": "16-mixed"},
        "log_config": {"name": "ldm_small", "save_dir": str(tmp_path / "logs"), "check_dir": str(tmp_path / "logs" / "check"),
                       "checkpoint": True, "model_summary": False},
    }
    path = tmp_path / "ldm.yaml"
    path.write_text(yaml.safe_dump(cfg, sort_keys=False))
    model, logger = run.main(["--config", str(path), "--synthetic"])
    out = capsys.readouterr().out
    assert isinstance(model, vision_amd.LatentDiffusion) and model.first_stage is not None and "TeL" in out
    v0 = tmp_path / "logs" / "ldm_small" / "version_0"
    rows = [json.loads(l) for l in open(v0 / "metrics.jsonl")]
    assert len(rows) == 2 and all(0.0 < r["TL"] < 10.0 and 0.0 < r["VL"] < 10.0 for r in rows)
    from PIL import Image
    for e in (0, 1):
        assert Image.open(v0 / "Samples" / f"ldm_small_Epoch_{e}.png").size == (2 * (16 + 2) + 2, 2 * (16 + 2) + 2)
    last = load_file(str(v0 / "checkpoints" / "last.safetensors"))
    assert set(last) == set(model.state_dict()) and "linear_1.linear.weight" in last and not any("vq_layer" in k for k in last)


@pytest.mark.parametrize("preset", ["vqvaeunet", "ldm", "flowers"])
def test_bench_presets_print_one_json_line(preset):
    """bench.py on the first-stage / latent-diffusion presets and on a small batch of the flowers DDPM (two steps, per-layer
    table on): one JSON line with the contract's keys, a finite positive rate, the roofline object with the fields it is
    computed from, and the exchange object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    extra = ["--batch", "2"] if preset == "flowers" else []
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--preset", preset, "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"] + extra, cwd=root, capture_output=True, text=True, timeout=200,
                       env=dict(os.environ, VAMD_BENCH_LAYERS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["unit"] == "images/s" and line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 2
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0 and 0 < roof["frac"] < 1
    assert roof["launches"] > 0 and roof["avg_launch_ms"] > 0 and roof["flops_per_launch"] > 0
    # achieved = FLOPs per launch / average launch time, frac = achieved / peak: the line must be consistent with itself
    ach = roof["flops_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e12
    assert abs(ach - roof["achieved"]) <= 0.02 * roof["achieved"] + 0.1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert 0 < roof["executed_frac"] <= roof["frac"] + 1e-4
    assert line["wgrad_kernel"]["launches"] > 0 and line["wgrad_kernel"]["achieved_tflops"] > 0
    assert line["host_issue_ms_per_step"] > 0 and line["mfma_conv_ms_per_step"] > 0
    assert line["exchange"]["ranks"] == 1 and not line["exchange"]["live"]
    assert ("sampling" in line) and (line["sampling"] is None) == (preset == "vqvaeunet")
    assert "layer (" in r.stderr


def test_bench_with_a_captured_step():
    """bench.py --graph: the timed steps are replays of one captured training step; the line says so, the host issues a
    step in a fraction of a millisecond, and the roofline object (from the instrumented steps after the timed ones) is still there."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--preset", "flowers", "--batch", "2", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline", "--no-sampling", "--graph"], cwd=root, capture_output=True,
                       text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["graph_replay"] is True and line["value"] > 0 and line["steps"] == 4
    assert 0 < line["host_issue_ms_per_step"] < 2.0, line["host_issue_ms_per_step"]
    assert line["roofline"]["launches"] > 0 and "per-launch HIP events" in line["roofline"]["measured_over"]
    assert 0 < line["final_loss"] < 5
