"""End-to-end parity of the HIP DDPMModel against golden vectors captured from the unmodified
reference (tests/golden/*.safetensors): forward, loss, every parameter gradient, BatchNorm running
statistics, three AdamW steps, validation loss and the fixed-noise sampler.

How gradients are judged.  Two properties of the reference function limit what any
implementation can reproduce:
  * the sinusoidal time embedding evaluates sin/cos of 2*pi*1000*n^2 (about 6e3 rad), so a 1-ulp
    difference in the float32 noise rate n (device sinf vs host sin) moves the embedding by ~5e-4;
  * the loss mean|pred - eps| has the discontinuous gradient sign(pred - eps)/N, and several
    parameter gradients are sums with heavy cancellation.
`test_backward_exact_inputs` therefore feeds the network the reference's own x_t and n^2 bits
(through forward(x, t)) and compares every parameter gradient with the exact gradient of the same
function, evaluated in fp64 by the oracle (relative Frobenius norm per tensor):
  fp32 compute: error <= max(2e-5, 4 x the error of the reference's own fp32 gradient), per tensor
                and (with 1e-5 as the floor) over all parameters together
  bf16 compute: error <= max(8e-2, 3 x the error of a bf16-autocast run of the reference
                arithmetic), and <= max(8e-2, 2 x that figure) over all parameters together.
`test_loss_grads_bnstats` runs the whole training objective p_loss (schedule evaluated on the
device) and checks the loss (1e-5 / 1e-2 relative), the BatchNorm statistics and the gradients
over all parameters together (2e-2 fp32; as above for bf16).
Parameters whose exact gradient is zero (biases feeding a train-mode BatchNorm, the key bias of
the softmax attention) must only stay negligible.
Other tolerances: forward 1e-4 / 2e-2 vs the golden output, sampler 1e-3 / 5e-2 of max|ref| per
pixel (fp32 / bf16).
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import vision_amd                         # noqa: E402
from vision_amd import _lib               # noqa: E402
from conftest import noise_keys, sub      # noqa: E402
from oracle import restatement as R       # noqa: E402  (checker only)

DEV = torch.device("cuda:0")
MODES = {
    "f32-direct": (torch.float32, _lib.ALGO_DIRECT),
    "f32-auto": (torch.float32, _lib.ALGO_AUTO),
    "bf16-direct": (torch.bfloat16, _lib.ALGO_DIRECT),
    "bf16-auto": (torch.bfloat16, _lib.ALGO_AUTO),
}
TOL = {
    torch.float32: dict(fwd=1e-4, loss=1e-5, grad=5e-4, bn=1e-4, state=5e-4, gen=1e-3),
    torch.bfloat16: dict(fwd=2e-2, loss=1e-2, grad=6e-2, bn=1e-2, state=2e-2, gen=5e-2),
}


# three AdamW steps, fp32 compute, update of every parameter tensor against the reference's: over all elements, over the
# elements whose update has the reference's direction, and the share of elements that moved the other way
ADAM_REL_ALL, ADAM_REL_KEPT, ADAM_FLIPPED = 0.4, 0.4, 0.2
# ... and of all parameters as one vector: measured 0.005 - 0.085 on the four fixtures, bound = twice the largest
ADAM_REL_GLOBAL = 0.17


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def make_model(cfg, state, mode):
    dtype, algo = MODES[mode]
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=dtype, kernel_algo=algo)
    m.load_state_dict(state)
    return m.to(DEV)


def masks_for(g, cfg, dtype, prefix=""):
    if cfg["dropout"] == 0.0:
        return None
    return {name: g[f"{prefix}dropmask/{i}"].permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
            for i, name in enumerate(("Middle.0", "Middle.2"))}


@pytest.fixture(scope="module", params=list(MODES))
def mode(request):
    return request.param


def test_forward_eval(golden, mode):
    name, g, cfg, B = golden
    tol = TOL[MODES[mode][0]]
    m = make_model(cfg, sub(g, "state0"), mode).eval()
    with torch.no_grad():
        out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))
    assert out.shape == g["fwd_eval"].shape and out.dtype == torch.float32
    assert rel(out, g["fwd_eval"]) < tol["fwd"]
    n, s = m.diffusion_schedule(g["t"].to(DEV))
    assert (n.cpu() - g["sched_n"]).abs().max() < 3e-7 and (s.cpu() - g["sched_s"]).abs().max() < 3e-7


_TRUTH = {}


def _forward_l1_grads(sd, cfg, x_t, n2, eps, masks, dtype):
    """Gradient of mean|unet(x_t, n2) - eps| for every parameter, evaluated by the oracle."""
    cast = (lambda v: v.to(dtype)) if dtype is not None else (lambda v: v)
    leaves = {k: cast(v).detach().clone().requires_grad_(True) for k, v in sd.items() if R.is_param(k)}
    work = {k: (cast(v).clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}   # never alias the golden state
    work.update(leaves)
    m = None if masks is None else {k: cast(v) for k, v in masks.items()}
    pred = R.unet_forward(work, cfg, cast(x_t), cast(n2), True, m)
    loss = torch.nn.functional.l1_loss(pred.float() if dtype is None else pred, cast(eps) if dtype is not None else eps)
    gs = torch.autograd.grad(loss, list(leaves.values()))
    return pred.detach(), dict(zip(leaves, gs))


def exact_gradients(name, g, cfg):
    """(exact fp64 prediction and gradients for the reference's x_t / n^2, per-tensor error of the
    reference fp32 gradients, per-tensor and global error of a bf16-autocast run), cached."""
    if name not in _TRUTH:
        sd = sub(g, "state0")
        masks = None
        if cfg["dropout"] != 0.0:
            masks = {"Middle.0": g["dropmask/0"], "Middle.2": g["dropmask/1"]}
        n2 = g["sched_n"] ** 2
        pred64, g64 = _forward_l1_grads(sd, cfg, g["x_t"], n2, g["eps"], masks, torch.float64)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            _, gbf = _forward_l1_grads(sd, cfg, g["x_t"], n2, g["eps"], masks, None)
        ref = sub(g, "grad")
        zero = noise_keys(g)
        e32 = {k: rel(ref[k], g64[k]) for k in ref if k not in zero}
        ebf = {k: rel(gbf[k], g64[k]) for k in ref if k not in zero}
        keys = sorted(e32)
        cat = lambda d: torch.cat([d[k].double().flatten() for k in keys])
        _TRUTH[name] = (pred64, g64, e32, ebf, rel(cat(gbf), cat(g64)), rel(cat(ref), cat(g64)))
    return _TRUTH[name]


def _cat(d, keys):
    return torch.cat([d[k].detach().double().flatten().cpu() for k in keys])


def test_backward_exact_inputs(golden, mode):
    """forward(x, t) on the reference's x_t / n^2 bits, L1 against eps, every gradient vs exact."""
    name, g, cfg, B = golden
    dtype = MODES[mode][0]
    pred64, g64, e32, ebf, ebf_all, e32_all = exact_gradients(name, g, cfg)
    m = make_model(cfg, sub(g, "state0"), mode).train()
    m.set_dropout_masks(masks_for(g, cfg, dtype))
    out = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV))
    assert rel(out, pred64) < (1e-5 if dtype == torch.float32 else 2e-2)
    loss = torch.nn.functional.l1_loss(out, g["eps"].to(DEV))
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < TOL[dtype]["loss"] * abs(g["loss"].item())
    grads = {k: p.grad for k, p in m.named_parameters()}
    scale = max(v.abs().max().item() for v in g64.values())
    failures = []
    for k in g64:
        assert grads[k] is not None and torch.isfinite(grads[k]).all(), k
        if k not in e32:                      # exact gradient is zero
            if grads[k].abs().max().item() > (1e-4 if dtype == torch.float32 else 5e-2) * scale:
                failures.append((k, "nonzero", grads[k].abs().max().item()))
            continue
        e = rel(grads[k], g64[k])
        bound = max(2e-5, 4 * e32[k]) if dtype == torch.float32 else max(8e-2, 3 * ebf[k])
        if e > bound:
            failures.append((k, e, bound))
    assert not failures, failures
    keys = sorted(e32)
    e_all = rel(_cat(grads, keys), _cat(g64, keys))
    # a float32 forward may round a few pred - eps differences to the other side of zero than the
    # exact one does (the reference's own fp32 run does): allow what the reference itself shows
    # bf16: measured 0.030-0.058 over the four configurations against 0.027-0.064 for the reference's own bf16
    # autocast on the same inputs (tests/diag_bf16_bounds_gpu.py): the bound is 1.5 x the reference's error, which is
    # 1.1-1.7 x what the engine shows; per tensor the engine reaches at most 2.3 x the reference's error (bound: 3 x)
    assert e_all < (max(1e-5, 4 * e32_all) if dtype == torch.float32 else 1.5 * ebf_all), e_all


def test_loss_grads_bnstats(golden, mode):
    """The whole training objective p_loss (schedule + noising on the device) vs the golden step."""
    name, g, cfg, B = golden
    dtype = MODES[mode][0]
    tol = TOL[dtype]
    _, g64, e32, ebf, ebf_all, _e32_all = exact_gradients(name, g, cfg)
    m = make_model(cfg, sub(g, "state0"), mode).train()
    m.set_dropout_masks(masks_for(g, cfg, dtype))
    loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) < tol["loss"] * abs(g["loss"].item())
    grads = {k: p.grad for k, p in m.named_parameters()}
    ref = sub(g, "grad")
    keys = sorted(e32)
    e_all = rel(_cat(grads, keys), _cat(ref, keys))
    print(f"p_loss gradients vs the reference's fp32 gradients, {name} {mode}: {e_all:.2e} (reference bf16 autocast: {ebf_all:.2e})")
    # measured: fp32 2.4e-5 ... 1.3e-3 (the larger ones are single L1 sign flips reached through the device-side schedule);
    # bf16 at most 1.35 x the reference's own bf16 autocast error on the same step
    assert e_all < (2.6e-3 if dtype == torch.float32 else 1.7 * ebf_all), e_all
    sd = m.state_dict()
    for k, v in sub(g, "bn1").items():
        if v.dtype == torch.int64:
            assert int(sd[k]) == int(v), k
        else:
            assert rel(sd[k], v) < tol["bn"], k


def test_three_adamw_steps_then_eval_and_sample(golden, mode):
    name, g, cfg, B = golden
    dtype = MODES[mode][0]
    tol = TOL[dtype]
    m = make_model(cfg, sub(g, "state0"), mode).train()
    opts, scheds = m.configure_optimizers()
    opt = opts[0]
    x0 = g["x0"].to(DEV)
    losses = []
    for step in range(3):
        torch.manual_seed(100 + step)                       # CPU generator: the reference's draws
        eps = torch.randn_like(g["x0"])
        t = torch.rand((B, 1, 1, 1))
        m.set_dropout_masks(masks_for(g, cfg, dtype, f"step{step}/"))
        loss = m.p_loss(x0, eps.to(DEV), t.to(DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    ref_losses = g["loss_steps"].tolist()
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < max(tol["loss"], 2e-4) * abs(b) * (1 if dtype == torch.float32 else 3), (losses, ref_losses)
    # Adam divides every element's update by its own gradient magnitude, so an element whose tiny
    # gradient changes sign under different rounding moves by 2*lr per step: parameters are checked
    # against the per-element Adam bound everywhere and, in fp32, through the bulk of the update.
    sd = m.state_dict()
    p0 = sub(g, "state0")
    lr = cfg["learning_rate"]
    all_du, all_dv = [], []
    for k, v in sub(g, "state3").items():
        if v.dtype == torch.int64:
            assert int(sd[k]) == int(v), k
        elif "running_" in k:
            assert rel(sd[k], v) < (2e-2 if dtype == torch.float32 else 1e-1), k   # after Adam's sign-sensitive steps
        else:
            assert (sd[k].cpu() - v).abs().max().item() <= 2 * 3 * lr * 1.05 + 1e-6, k
            if dtype == torch.float32 and k not in noise_keys(g):
                # 0.4: the schedule's sin / cos differ from torch's CPU ones in the last bit, the highest embedding
                # frequency (2 pi 1000) turns that into ~1e-4 on the prediction, and one L1 sign flip among the 3072
                # outputs moves the gradients it reaches by ~2 / sqrt(3072) (seen: 0.31 on one BatchNorm weight of
                # t0_pool, whose forward-path gradients agree with fp64 to 5e-6)
                du, dv = sd[k].cpu() - p0[k], v - p0[k]
                flipped = (du * dv) < 0            # elements that moved the other way: an L1 / Adam sign flip
                r_all, r_kept = rel(du, dv), rel(du[~flipped], dv[~flipped])
                if os.environ.get("VAMD_TEST_REPORT"):
                    with open(os.environ["VAMD_TEST_REPORT"], "a") as f:
                        f.write(f"{mode} {k} numel {v.numel()} flipped {int(flipped.sum())} rel_all {r_all:.4f} rel_kept {r_kept:.4f}\n")
                assert r_all < ADAM_REL_ALL and r_kept < ADAM_REL_KEPT and flipped.float().mean().item() <= ADAM_FLIPPED, \
                    (k, r_all, r_kept, int(flipped.sum()), v.numel())
                all_du.append(du.flatten().double())
                all_dv.append(dv.flatten().double())
    if all_du:
        # the three-step update of the whole model (every tensor above in one vector): the single BatchNorm tensors that
        # reach 0.16 - 0.31 hold 32 - 64 elements each; over all parameters the update agrees to ADAM_REL_GLOBAL
        du, dv = torch.cat(all_du), torch.cat(all_dv)
        r_glob = ((du - dv).norm() / dv.norm()).item()
        if os.environ.get("VAMD_TEST_REPORT"):
            with open(os.environ["VAMD_TEST_REPORT"], "a") as f:
                f.write(f"{mode} GLOBAL numel {dv.numel()} rel_all {r_glob:.5f}\n")
        assert r_glob < ADAM_REL_GLOBAL, r_glob

    # validation loss + sampler from the REFERENCE's trained state (isolates them from step drift)
    m2 = make_model(cfg, sub(g, "state3"), mode).eval()
    with torch.no_grad():
        torch.manual_seed(200)
        eps = torch.randn_like(g["x0"])
        t = torch.rand((B, 1, 1, 1))
        val = m2.p_loss(x0, eps.to(DEV), t.to(DEV))
        assert abs(val.item() - g["val_loss"].item()) < tol["loss"] * abs(g["val_loss"].item())
        for T in (1, 4):
            out = m2.generate(2, T, initial_noise=g["gen_noise"].to(DEV))
            ref = g[f"gen_T{T}"]
            assert (out.cpu() - ref).abs().max().item() < tol["gen"] * ref.abs().max().item(), T
        # batch-sharded sampling == unsharded, row for row (eval BatchNorm uses running statistics)
        a = m2.generate(1, 4, initial_noise=g["gen_noise"][:1].to(DEV))
        b = m2.generate(1, 4, initial_noise=g["gen_noise"][1:].to(DEV))
        full = m2.generate(2, 4, initial_noise=g["gen_noise"].to(DEV))
        assert (torch.cat([a, b]) - full).abs().max().item() <= 1e-5 * full.abs().max().item()


def test_state_dict_roundtrip_and_checkpoint(golden, mode, tmp_path):
    name, g, cfg, B = golden
    if mode != "bf16-auto":
        pytest.skip("layout-independent; one mode is enough")
    from safetensors.torch import load_file, save_file
    m = make_model(cfg, sub(g, "state3"), mode)
    m._ensure_engine()
    sd = {k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()}
    for k, v in sub(g, "state3").items():
        assert torch.equal(sd[k], v), k
    path = str(tmp_path / "last.safetensors")
    save_file(sd, path)
    m3 = vision_amd.DDPMModel.load_from_checkpoint(path, cfg)
    for k, v in load_file(path).items():
        assert torch.equal(m3.state_dict()[k], v), k


def test_eval_batchnorm_coefficients_follow_state_changes(golden, mode):
    """Eval-mode BatchNorm coefficients are cached between forwards (a sampler's steps): a new state, an
    in-place edit of a running statistic and a training step must each show in the next eval forward."""
    name, g, cfg, B = golden
    if mode != "bf16-auto":
        pytest.skip("host logic; one mode is enough")
    x, t2 = g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV)
    m = make_model(cfg, sub(g, "state0"), mode).eval()
    fresh3 = make_model(cfg, sub(g, "state3"), mode).eval()
    with torch.no_grad():
        y0 = m(x, t2)
        assert torch.equal(m(x, t2), y0)                         # cached coefficients: same result
        want3 = fresh3(x, t2)
        m.load_state_dict(sub(g, "state3"))                      # other parameters AND running statistics
        assert torch.equal(m(x, t2), want3)
        key = next(k for k in m.state_dict() if k.endswith("running_var"))
        m.state_dict()[key].mul_(4.0)                            # in-place edit of a buffer
        y_edit = m(x, t2)
        assert not torch.equal(y_edit, want3)
        fresh3.state_dict()[key].mul_(4.0)
        assert torch.equal(fresh3(x, t2), y_edit)
    m.train()
    loss = m.p_loss(g["x0"].to(DEV), g["eps"].to(DEV), g["t"].to(DEV))             # updates the running statistics
    loss.backward()
    m.eval()
    with torch.no_grad():
        assert not torch.equal(m(x, t2), y_edit)
