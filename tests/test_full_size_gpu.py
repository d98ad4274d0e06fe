"""Parity at the benchmark's full sizes (butterfly_ddpm: 224x224, batch 64, bf16), where the CPU
oracle cannot evaluate the whole batch in seconds:

  * every distinct convolution shape of the UNet at batch 64, through the C ABI: the result for
    images [16, 32) must be bit-identical to a batch-16 call on those images alone (forward and input
    gradient: the per-output reduction order does not depend on the launch geometry), the weight
    and bias gradients must equal the sum of the four quarter-batch gradients, and the LAST image
    (the far end of every tensor) is checked against an fp64 CPU convolution;
  * the full-resolution model on 2 images against the oracle in fp64 (prediction, loss, global
    gradient error; tolerances in the test);
  * eval-mode forward at batch 64 equals the four batch-16 forwards bit for bit, and a training
    step repeated from the same state reproduces its loss and gradients.
"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import vision_amd                                    # noqa: E402
from vision_amd import _lib                          # noqa: E402
from vision_amd._lib import ConvDesc, call, lib      # noqa: E402
from vision_amd.presets import get_preset            # noqa: E402

from test_kernels_gpu import S, relerr, pack_weights   # noqa: E402

DEV = torch.device("cuda:0")
BF, DT = torch.bfloat16, _lib.BF16
_KEEP = []


def P(t):
    """Device pointer of t, kept alive until the test ends (the kernels are asynchronous)."""
    if t is None:
        return None
    _KEEP.append(t)
    return t.data_ptr()


@pytest.fixture(autouse=True)
def _release_temporaries():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()
    torch.cuda.empty_cache()

# B, H, W, Cin, Cout, k, stride, up: the distinct layer shapes of the butterfly / flowers UNet
FULL_CASES = [
    (64, 224, 224, 3, 64, 3, 1, 0),
    (64, 224, 224, 3, 64, 1, 1, 0),
    (64, 224, 224, 64, 64, 3, 1, 0),
    (64, 224, 224, 64, 64, 3, 2, 0),
    (64, 112, 112, 64, 128, 3, 1, 0),
    (64, 112, 112, 64, 128, 1, 1, 0),
    (64, 112, 112, 128, 128, 3, 1, 0),
    (64, 112, 112, 128, 128, 3, 2, 0),
    (64, 56, 56, 128, 256, 3, 1, 0),
    (64, 56, 56, 256, 256, 3, 1, 0),
    (64, 28, 28, 512, 512, 3, 1, 0),
    (64, 14, 14, 512, 512, 3, 1, 1),
    (64, 14, 14, 512, 512, 1, 1, 0),
    (64, 56, 56, 128, 128, 3, 1, 1),
    (64, 112, 112, 64, 64, 3, 1, 1),
    (64, 112, 112, 128, 64, 3, 1, 0),
    (64, 224, 224, 64, 3, 3, 1, 0),
    (64, 224, 224, 64, 3, 1, 1, 0),
    (64, 224, 224, 3, 3, 3, 1, 0),
]


def _run(case, B, x, wf, wd, dy, bias, with_dbias=True):
    """fwd, dgrad, wgrad (+dbias) of `case` with batch B on NHWC tensors x / dy."""
    _, H, W, Cin, Cout, k, stride, up = case
    d = ConvDesc(B, H, W, Cin, Cout, k, stride, 1 if k == 3 else 0, up, DT, 0)
    Ho, Wo = dy.shape[1], dy.shape[2]
    y = torch.empty((B, Ho, Wo, Cout), dtype=BF, device=DEV)
    call("vamd_conv2d_fwd", C.byref(d), P(x), P(wf), P(bias), None, None, None, P(y), S())
    dx = torch.empty((B, H, W, Cin), dtype=BF, device=DEV)
    nb = lib.vamd_conv2d_dgrad_workspace(C.byref(d))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    call("vamd_conv2d_dgrad", C.byref(d), P(dy), P(wd), None, P(dx), P(ws), nb, S())
    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=DEV)
    db = torch.empty((Cout,), dtype=torch.float32, device=DEV)
    nb2 = lib.vamd_conv2d_wgrad_workspace(C.byref(d))
    ws2 = torch.empty(max(nb2, 16), dtype=torch.uint8, device=DEV)
    call("vamd_conv2d_wgrad_oihw", C.byref(d), P(dy), P(x), P(dw), P(db), P(ws2), nb2, S())
    return y, dx, dw, db


@pytest.mark.parametrize("case", FULL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_full_size_conv_layers(case, no_splitk):
    B, H, W, Cin, Cout, k, stride, up = case
    gen = torch.Generator().manual_seed(11)
    pad = 1 if k == 3 else 0
    He, We = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (He + 2 * pad - k) // stride + 1, (We + 2 * pad - k) // stride + 1
    dgen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn((B, H, W, Cin), device=DEV, generator=dgen).to(BF)
    dy = torch.randn((B, Ho, Wo, Cout), device=DEV, generator=dgen).to(BF)
    w64 = (torch.randn((Cout, Cin, k, k), generator=gen) / math.sqrt(Cin * k * k)).to(BF).double()
    bias = torch.randn(Cout, generator=gen).to(DEV)
    wf, wd = pack_weights(w64, BF, DT)

    y, dx, dw, db = _run(case, B, x, wf, wd, dy, bias)

    # quarter batches: bit-identical outputs, gradients of the parameters add up
    q = B // 4
    dw_sum = torch.zeros_like(dw, dtype=torch.float64)
    db_sum = torch.zeros_like(db, dtype=torch.float64)
    for i in range(4):
        sl = slice(i * q, (i + 1) * q)
        yq, dxq, dwq, dbq = _run(case, q, x[sl], wf, wd, dy[sl], bias)
        assert torch.equal(yq, y[sl]), f"forward of images {i * q}..{(i + 1) * q} depends on the batch"
        assert torch.equal(dxq, dx[sl]), f"input gradient of images {i * q}..{(i + 1) * q} depends on the batch"
        dw_sum += dwq.double()
        db_sum += dbq.double()
    # fp32 sums in a different association; the thin-channel layers add their bias-gradient block
    # partials with atomics, so that one also moves from run to run
    assert relerr(dw, dw_sum) < 1e-5
    assert relerr(db, db_sum) < 1e-5

    # the last image against an fp64 CPU convolution
    xl = x[-1:].permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    wl = w64.clone().requires_grad_(True)
    xin = F.interpolate(xl, scale_factor=2, mode="nearest") if up else xl
    yl = F.conv2d(xin, wl, bias.double().cpu(), stride=stride, padding=pad)
    dyl = dy[-1:].permute(0, 3, 1, 2).double().cpu()
    gx, gw = torch.autograd.grad(yl, [xl, wl], dyl)
    assert relerr(y[-1:].permute(0, 3, 1, 2), yl.detach()) < 4e-3
    assert relerr(dx[-1:].permute(0, 3, 1, 2), gx) < 4e-3
    _, _, dw1, db1 = _run(case, 1, x[-1:], wf, wd, dy[-1:], bias)
    assert relerr(dw1, gw) < 2e-4
    assert relerr(db1, dyl.sum((0, 2, 3))) < 2e-4


def _butterfly_model(compute_dtype):
    from oracle import restatement as R
    # dropout off: its masks come from different generators in the oracle and on the GPU (the
    # masked path is pinned by the t0_drop golden vectors)
    cfg = dict(get_preset("butterfly")["model_config"], dropout=0.0)
    sd = R.init_state_like_torch(cfg, cfg["manual_seed"])
    model = vision_amd.get_model("ddpm", cfg, compute_dtype=compute_dtype).to(DEV)
    model.load_state_dict(sd)
    return model, cfg, sd, R


@pytest.mark.parametrize("fuse_shortcut", [False, True], ids=["two-launch-shortcut", "shortcut-in-conv2"])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_full_resolution_model_against_oracle(mode, fuse_shortcut, monkeypatch):
    """224x224 butterfly UNet on 2 images against the oracle evaluated in fp64.
    Train-mode prediction: 2e-6 (fp32 compute; torch's own fp32 forward is at 2.4e-7) / 2e-2 (bf16).
    Loss: 1e-5 / 2e-2.  Gradients, global relative error: 1.4e-3 / 2.2e-2 (twice what is measured) - the L1 loss's sign() turns
    a forward rounding difference of 1e-6 into flipped gradient signs for the outputs that sit
    that close to their target, which costs ~1e-3 of the gradient norm (the oracle's own fp32
    gradient is 7e-5 away from its fp64 gradient); measured 7e-4 / 1.1e-2."""
    import vision_amd.engine as engine_mod
    monkeypatch.setattr(engine_mod, "FUSE_SHORTCUT", fuse_shortcut)      # opt-in path of ResNetBlock's conv2
    model, cfg, sd, R = _butterfly_model(mode)
    model.train()
    P_ = cfg["patch_size"]
    g = torch.Generator().manual_seed(3)
    x = torch.rand((2, 3, P_, P_), generator=g)
    eps = torch.randn((2, 3, P_, P_), generator=g)
    t = torch.rand((2, 1, 1, 1), generator=g)
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    with torch.no_grad():
        pred64 = R.unet_forward({k: v.clone() for k, v in sd64.items()}, cfg, eps.double(), t.double(), training=True)
        bufs = model._ensure_engine().flat_b.clone()
        pred = model(eps.to(DEV), t.to(DEV)).double().cpu()
        model._ensure_engine().flat_b.copy_(bufs)          # undo the running-statistics update
    assert relerr(pred, pred64) < (2e-6 if mode == "fp32" else 2e-2)

    ref_loss, ref64 = R.loss_and_grads(sd64, cfg, x.double(), eps.double(), t.double())
    loss = model.p_loss(x.to(DEV), noises=eps.to(DEV), diffusion_times=t.to(DEV))
    loss.backward()
    tol = 1e-5 if mode == "fp32" else 2e-2
    assert abs(float(loss.detach()) - float(ref_loss)) < tol * abs(float(ref_loss)), (float(loss.detach()), float(ref_loss))
    names = [n for n, _ in model.named_parameters()]
    want = torch.cat([ref64[n].flatten() for n in names])
    got = torch.cat([p.grad.detach().flatten().double().cpu() for _, p in model.named_parameters()])
    err = ((got - want).norm() / want.norm()).item()
    print(f"global gradient error, 224x224 model, {mode}: {err:.2e}")
    assert err < (1.4e-3 if mode == "fp32" else 2.2e-2), err


def test_full_batch_forward_is_batch_independent_and_step_is_reproducible(no_splitk):
    model, cfg, _, _ = _butterfly_model("bf16")
    P_, B = cfg["patch_size"], 64
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((B, 3, P_, P_), device=DEV, generator=gen)
    t = torch.rand((B, 1, 1, 1), device=DEV, generator=gen)
    model.eval()
    with torch.no_grad():
        full = model(x, t)
        for i in range(4):
            sl = slice(16 * i, 16 * (i + 1))
            assert torch.equal(model(x[sl], t[sl]), full[sl]), f"images {16 * i}.. depend on their batch"
    assert torch.isfinite(full).all()

    model.train()
    eng = model._ensure_engine()
    img = torch.rand((B, 3, P_, P_), device=DEV, generator=gen)
    eps = torch.randn((B, 3, P_, P_), device=DEV, generator=gen)
    tt = torch.rand((B, 1, 1, 1), device=DEV, generator=gen)
    bufs = eng.flat_b.clone()
    runs = []
    for _ in range(2):
        eng.flat_b.copy_(bufs)                         # same BatchNorm running statistics
        eng.flat_g.zero_()
        loss = model.p_loss(img, noises=eps, diffusion_times=tt)
        loss.backward()
        runs.append((loss.detach().clone(), eng.flat_g.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    # everything on the path adds in a fixed order except the column sums of the thin-channel
    # layers' bias gradients, which use fp32 atomics
    err = ((runs[0][1] - runs[1][1]).norm() / runs[0][1].norm()).item()
    assert err < 1e-6, err


def test_in_kernel_input_normalisation_changes_no_bit_of_a_training_step(monkeypatch):
    """The opt-in form of BatchNorm + LeakyReLU in front of the 64-channel 3 x 3 convolutions (normalised inside the
    forward and weight-gradient kernels, no normalised tensor stored) against the default form: same loss, same
    gradients, same running statistics, bit for bit, on a 16-image 224 x 224 training step (conv2 of the first block,
    conv1 of the third and conv2 of the sixth block take that path)."""
    import vision_amd.engine as engine_mod
    gen = torch.Generator(device=DEV).manual_seed(11)
    B = 16
    img = torch.rand((B, 3, 224, 224), device=DEV, generator=gen)
    eps = torch.randn((B, 3, 224, 224), device=DEV, generator=gen)
    tt = torch.rand((B, 1, 1, 1), device=DEV, generator=gen)
    runs = []
    for fused in (False, True):
        monkeypatch.setattr(engine_mod, "FUSE_TRAIN_NORM", fused)
        model, cfg, _, _ = _butterfly_model("bf16")
        model.train()
        eng = model._ensure_engine()
        taken = [c.name for c in eng.all_convs() if c.k == 3 and c.plan(B, *_in_hw(eng, c)).prenorm]
        assert bool(taken) == fused, taken
        loss = model.p_loss(img, noises=eps, diffusion_times=tt)
        loss.backward()
        runs.append((loss.detach().clone(), eng.flat_g.clone(), eng.flat_b.clone()))
        del model, eng
    assert torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][2], runs[1][2])
    # (the thin-channel layers' bias gradients use fp32 atomics: compare everything else exactly)
    same = runs[0][1] == runs[1][1]
    assert same.float().mean().item() > 0.9999 and ((runs[0][1] - runs[1][1]).norm() / runs[0][1].norm()).item() < 1e-6


def _in_hw(eng, conv):
    """Input height and width of a convolution of the 224 x 224 butterfly UNet, from its position in the plan."""
    for layer in eng.layers:
        if conv in layer.convs:
            r = int(layer.spec.res_in)
            return r, r
    return 224, 224


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_full_resolution_sampler_against_oracle(mode):
    """Three sampler steps on two 224x224 images against the oracle (fp32, as the reference runs
    it): 1e-4 (fp32 compute) / 3e-2 (bf16 compute) relative error of the returned images."""
    model, cfg, sd, R = _butterfly_model(mode)
    model.eval()
    P_ = cfg["patch_size"]
    noise = torch.randn((2, 3, P_, P_), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        want = R.reverse_diffusion({k: v.clone() for k, v in sd.items()}, cfg, noise, 3)
    got = model.generate(2, 3, initial_noise=noise.to(DEV))
    assert got.shape == want.shape
    assert relerr(got, want) < (1e-4 if mode == "fp32" else 3e-2)
