"""The Lightning-level surface of the reference executed on the GPU: training / validation / test steps,
denoise, the epoch hooks, and run.py's whole flow (versioned log directory, metrics, loss file, sample
grids, top-k + last safetensors checkpoints, config copy, resume from the previous version).

Reference: models/ddpm.py:257-260 (denoise), :301-337 (steps and hooks), :354-368 (sample_images),
run.py:42-103, utils.py:47-95.  Golden values come from tests/golden (the unmodified reference).
The reference draws noise and times from torch's CPU generator; the steps here draw on the device, so
the tests that compare with golden losses substitute the reference's CPU draws for the device draws.
"""
import json
import os
from unittest import mock

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import vision_amd                          # noqa: E402
from conftest import ROOT, load_golden, sub  # noqa: E402

DEV = torch.device("cuda:0")


def _model(cfg, state, dtype=torch.float32):
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=dtype)
    m.load_state_dict(state)
    return m.to(DEV)


class _CpuDraws:
    """torch.randn_like / torch.rand that draw from the CPU generator (as the reference does on CPU
    tensors) and move the result to where the caller asked for it."""

    def __enter__(self):
        real_randn_like, real_rand = torch.randn_like, torch.rand

        def randn_like(t, **kw):
            return real_randn_like(torch.empty(t.shape, dtype=t.dtype)).to(t.device)

        def rand(*size, device=None, **kw):
            return real_rand(*size, **kw).to(device if device is not None else "cpu")

        self._p = [mock.patch.object(torch, "randn_like", randn_like), mock.patch.object(torch, "rand", rand)]
        for p in self._p:
            p.start()
        return self

    def __exit__(self, *exc):
        for p in self._p:
            p.stop()


def _masks(g, cfg, prefix):
    if cfg["dropout"] == 0.0:
        return None
    return {name: g[f"{prefix}dropmask/{i}"].permute(0, 2, 3, 1).contiguous().to(DEV)
            for i, name in enumerate(("Middle.0", "Middle.2"))}


def test_denoise_matches_golden(golden):
    name, g, cfg, B = golden
    m = _model(cfg, sub(g, "state0")).eval()
    with torch.no_grad():
        n, s = g["sched_n"].to(DEV), g["sched_s"].to(DEV)
        pred_noises, pred_images = m.denoise(g["x_t"].to(DEV), n, s)
    ref = g["denoise_pred_images"]
    assert (pred_noises.cpu() - g["fwd_eval"]).abs().max() <= 1e-4 * g["fwd_eval"].abs().max()
    assert (pred_images.cpu() - ref).abs().max() <= 1e-4 * ref.abs().max()


def test_training_validation_test_steps_match_golden(golden):
    """training_step on the reference's own draws == golden first-step loss; validation_step / test_step in
    eval mode under no_grad == golden validation loss; the logged names are TL / VL / TeL."""
    name, g, cfg, B = golden
    x0, y = g["x0"].to(DEV), torch.zeros(B, dtype=torch.int64, device=DEV)
    m = _model(cfg, sub(g, "state0")).train()
    m.set_dropout_masks(_masks(g, cfg, "step0/"))
    with _CpuDraws():
        torch.manual_seed(100)
        loss = m.training_step((x0, y), 0)
    ref = float(g["loss_steps"][0])
    assert abs(float(loss) - ref) < 2e-4 * abs(ref)
    assert loss.requires_grad and len(m.train_losses) == 1 and "TL" in m.logged
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())

    m2 = _model(cfg, sub(g, "state3")).eval()
    with torch.no_grad(), _CpuDraws():
        torch.manual_seed(200)
        assert m2.validation_step((x0, y), 0) is None
        torch.manual_seed(200)
        assert m2.test_step((x0, y), 0) is None
    ref = float(g["val_loss"])
    assert abs(float(m2.logged["VL"]) - ref) < 1e-4 * abs(ref)
    assert abs(float(m2.logged["TeL"]) - ref) < 1e-4 * abs(ref)
    assert len(m2.val_losses) == 1


def test_training_step_equals_p_loss_with_device_draws(golden):
    """With the device generator seeded alike, training_step draws what p_loss draws (randn_like, then
    rand: reference models/ddpm.py:263-264) and returns the same loss bit for bit."""
    name, g, cfg, B = golden
    if cfg["dropout"] != 0.0:
        pytest.skip("dropout masks come from a third draw; covered by the golden-mask tests")
    x0, y = g["x0"].to(DEV), torch.zeros(B, dtype=torch.int64, device=DEV)
    m = _model(cfg, sub(g, "state0"), torch.bfloat16).train()
    torch.manual_seed(5)
    a = m.training_step((x0, y), 0)
    sd = sub(g, "state0")
    m.load_state_dict(sd)                                   # undo the running-statistics update
    torch.manual_seed(5)
    b = m.p_loss(x0)
    assert torch.equal(a.detach(), b.detach())
    torch.manual_seed(5)
    eps = torch.randn_like(x0)
    t = torch.rand((B, 1, 1, 1), device=DEV)
    m.load_state_dict(sd)
    assert torch.equal(m.p_loss(x0, eps, t).detach(), b.detach())


def test_eval_mode_gradients_use_fixed_statistics(golden):
    """model.eval(); p_loss(x).backward(): BatchNorm normalises with the running statistics, which do not
    depend on the batch, so its input gradient has no mean / variance terms.  Checked against the oracle's
    autograd in fp64 on the same function (per-tensor relative error, global over all parameters)."""
    name, g, cfg, B = golden
    from oracle import restatement as R
    from test_model_gpu import _cat, rel
    sd = sub(g, "state3")                                   # running statistics away from their initial 0 / 1
    m = _model(cfg, sd).eval()
    loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))
    loss.backward()
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if R.is_param(k)}
    work = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    work.update(leaves)
    ref = R.p_loss(work, cfg, g["x0"].double(), g["eps"].double(), g["t"].double(), False)
    grads64 = dict(zip(leaves, torch.autograd.grad(ref, list(leaves.values()))))
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    got = {k: p.grad for k, p in m.named_parameters()}
    keys = sorted(k for k in grads64 if not k.endswith(".k_proj.bias"))     # exact gradient is zero there
    assert rel(_cat(got, keys), _cat(grads64, keys)) < 2e-2
    # conv biases in front of an eval-mode BatchNorm are NOT cancelled by a mean subtraction
    k1 = next(k for k in keys if k.endswith(".block1.0.2.bias"))
    assert float(grads64[k1].abs().max()) > 0 and rel(got[k1], grads64[k1]) < 5e-2


def _t1_config(tmp_path, max_epochs, name="t1_ddpm"):
    _, cfg, _ = load_golden("t1")
    return {
        "model_config": cfg,
        "data_config": {"patch_size": cfg["patch_size"], "data_dir": "unused", "batch_size": 4, "num_workers": 0,
                        "use_manual_split": True, "length": 60},
        "trainer_config": {"accelerator": "gpu", "devices": 1, "max_epochs": max_epochs, "precision": "16-mixed"},
        "log_config": {"name": name, "save_dir": str(tmp_path / "logs"), "check_dir": str(tmp_path / "logs" / "check"),
                       "checkpoint": True, "model_summary": False},
    }


def test_run_flow_checkpoints_and_resume(tmp_path, capsys):
    """run.py end to end on a small config: two epochs, then a second invocation that must start from
    version_0's last.safetensors (reference run.py:48-59)."""
    import run
    from safetensors.torch import load_file
    cfg_path = tmp_path / "t1.yaml"
    cfg_path.write_text(yaml.safe_dump(_t1_config(tmp_path, 2), sort_keys=False))
    model, logger = run.main(["--config", str(cfg_path), "--synthetic"])
    out = capsys.readouterr().out
    assert "No checkpoint found" in out and "TeL" in out
    v0 = tmp_path / "logs" / "t1_ddpm" / "version_0"
    assert logger.log_dir == str(v0)
    rows = [json.loads(l) for l in open(v0 / "metrics.jsonl")]
    assert len(rows) == 2 and all({"TL", "VL", "lr", "step"} <= set(r) for r in rows)
    lr0 = _t1_config(tmp_path, 2)["model_config"]["learning_rate"]
    assert abs(rows[0]["lr"] - lr0 * 0.99) < 1e-12 and abs(rows[1]["lr"] - lr0 * 0.99 ** 2) < 1e-12
    assert rows[0]["step"] == 12 and rows[1]["step"] == 24          # 48 training images / batch 4 per epoch
    assert all(np.isfinite([r["TL"], r["VL"]]).all() for r in rows)
    losses = np.load(v0 / "train_losses.npy")
    assert losses.shape == (2,) and abs(losses[0] - rows[0]["TL"]) < 1e-6 and abs(losses[1] - rows[1]["TL"]) < 1e-6
    for e in (0, 1):                                                  # sampling_period 1: a grid after every epoch
        from PIL import Image
        img = Image.open(v0 / "Samples" / f"t1_ddpm_Epoch_{e}.png")
        assert img.size == (2 * (32 + 2) + 2, 2 * (32 + 2) + 2)
    ck = sorted(os.listdir(v0 / "checkpoints"))
    tops = [f for f in ck if f.startswith("epoch=")]
    assert "last.safetensors" in ck and 1 <= len(tops) <= 2 and all("-VL=" in f for f in tops)
    assert yaml.safe_load(open(v0 / "config.yaml")) == _t1_config(tmp_path, 2)
    last = load_file(str(v0 / "checkpoints" / "last.safetensors"))
    final = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert set(last) == set(final) and all(torch.equal(last[k], final[k]) for k in last)
    assert int(last["Downsampling.0.block1.0.0.num_batches_tracked"]) == 24

    # second invocation, no epochs: the model it ends with is exactly what it loaded
    cfg_path.write_text(yaml.safe_dump(_t1_config(tmp_path, 0), sort_keys=False))
    model2, logger2 = run.main(["--config", str(cfg_path), "--synthetic"])
    out = capsys.readouterr().out
    assert logger2.version == 1 and f"Loaded checkpoint from {v0 / 'checkpoints' / 'last.safetensors'}" in out
    sd2 = {k: v.detach().cpu() for k, v in model2.state_dict().items()}
    assert all(torch.equal(last[k], sd2[k]) for k in last)


def test_default_config_path_runs_from_a_checkout(tmp_path):
    """`python run.py` without --config reads configs/flowers_ddpm.yaml, which the repository ships."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run.py"), "--synthetic", "--max-steps", "2",
                        "--save-dir", str(tmp_path / "logs")], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    v0 = tmp_path / "logs" / "flower_ddpm" / "version_0"
    assert (v0 / "metrics.jsonl").exists() and (v0 / "checkpoints" / "last.safetensors").exists()
    assert (v0 / "Samples" / "flower_ddpm_Epoch_0.png").exists()


def test_configure_optimizers_quirks(golden):
    name, g, cfg, B = golden
    from vision_amd.trainer import SyntheticImageFolder, Trainer
    m = _model(dict(cfg, scheduler_gamma=None), sub(g, "state0"))
    assert m.configure_optimizers() is None                 # reference models/ddpm.py:347-352
    with pytest.raises(ValueError, match="scheduler_gamma"):
        Trainer(max_epochs=1).fit(m, SyntheticImageFolder(cfg["patch_size"], 2, 20))
    bare = {k: v for k, v in cfg.items() if k != "scheduler_gamma"}
    opt = _model(bare, sub(g, "state0")).configure_optimizers()
    assert isinstance(opt, torch.optim.Optimizer)
    x = g["x_t"].to(DEV).requires_grad_(True)
    with pytest.raises(RuntimeError, match="parameter gradients only"):
        m(x, (g["sched_n"] ** 2).to(DEV))


def test_graph_replayed_steps_equal_eager_steps():
    """Training steps replayed from a HIP graph (vision_amd.graphed.GraphedTrainStep: two eager steps, one capture, replays)
    against the same steps issued launch by launch: identical losses, parameters, BatchNorm buffers and step counts after
    six steps on changing batches, with a learning rate that changes on the way (the optimiser reads it from the device)."""
    import vision_amd
    from vision_amd.graphed import GraphedTrainStep
    cfg = dict(name="ddpm", input_channels=3, learning_rate=1e-3, manual_seed=42, h_dims=[3, 32, 64, 64], down_index=[0, 1, 1, 1],
               attn_res=[8], with_conv=True, dropout=0.1, latent_dim=64, sampling_period=1, weight_decay=1e-3,
               diffusion_steps=4, scheduler_gamma=0.9, patch_size=32, grid=2)
    gen = torch.Generator().manual_seed(3)
    batches = [(torch.rand((4, 3, 32, 32), generator=gen).to(DEV), torch.zeros(4, dtype=torch.int64, device=DEV)) for _ in range(6)]
    out = []
    for graphed, on_device in ((False, True), (True, True)):
        torch.manual_seed(7)
        model = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to(DEV).train()
        (opt,), (sched,) = model.configure_optimizers()
        if on_device and not graphed:
            opt.make_capturable()                   # the same optimiser launch as the graph's: step count and rate on the device
        stepper = GraphedTrainStep(model, opt) if graphed else None
        torch.manual_seed(11)                       # noise, diffusion times and dropout masks of the six steps
        torch.cuda.manual_seed(11)
        losses = []
        for i, b in enumerate(batches):
            if graphed:
                losses.append(stepper(b, i))
            else:
                loss = model.training_step(b, i)
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
            if i == 3:
                sched.step()                        # the rate changes between two replays
        eng = model._ensure_engine()
        torch.cuda.synchronize()
        out.append((torch.stack(losses).cpu(), eng.flat_p.clone(), eng.flat_b.clone(), eng.flat_nbt.clone(), opt.state_steps(),
                    len(model.train_losses), torch.stack([l.detach() for l in model.train_losses]).cpu()))
    (l0, p0, b0, n0, s0, k0, t0), (l1, p1, b1, n1, s1, k1, t1) = out
    assert s0 == s1 == 6 and k0 == k1 == 6
    assert torch.equal(l0, l1) and torch.equal(t0, t1) and torch.equal(l1, t1)
    assert torch.equal(b0, b1) and torch.equal(n0, n1)
    assert torch.equal(p0, p1)


def test_weight_gradients_on_a_second_stream_change_no_bit():
    """Backward with its weight gradients on the engine's second stream (the default) against every kernel on one stream:
    identical gradients after one step, identical losses and parameters after four optimiser steps on changing batches;
    the second stream is joined when backward returns (nothing is left referenced for it)."""
    import vision_amd
    cfg = dict(name="ddpm", input_channels=3, learning_rate=1e-3, manual_seed=42, h_dims=[3, 32, 64, 64], down_index=[0, 1, 1, 1],
               attn_res=[8], with_conv=True, dropout=0.1, latent_dim=64, sampling_period=1, weight_decay=1e-3,
               diffusion_steps=4, scheduler_gamma=0.9, patch_size=32, grid=2)
    gen = torch.Generator().manual_seed(5)
    batches = [(torch.rand((4, 3, 32, 32), generator=gen).to(DEV), torch.zeros(4, dtype=torch.int64, device=DEV)) for _ in range(4)]
    out = []
    for two_streams in (True, False):
        torch.manual_seed(7)
        model = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to(DEV).train()
        (opt,), _ = model.configure_optimizers()
        eng = model._ensure_engine()
        eng.side_wgrad = two_streams
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        losses, first = [], None
        for i, b in enumerate(batches):
            loss = model.training_step(b, i)
            loss.backward()
            assert not eng._side_on and not eng._side_keep
            if first is None:
                torch.cuda.synchronize()
                first = eng.flat_g.clone()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        assert (eng._side_streams is not None) == two_streams
        out.append((first, torch.stack(losses).cpu(), eng.flat_p.clone()))
    (g0, l0, p0), (g1, l1, p1) = out
    assert torch.equal(g0, g1) and torch.equal(l0, l1) and torch.equal(p0, p1)
