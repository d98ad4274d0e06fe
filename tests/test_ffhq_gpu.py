"""The ffhq workload (reference configs/ffhq_ddpm.yaml:7-27: 1024x1024, six levels of widths
64/64/128/128/256/256, attention at 32x32 on the way down AND on the way up plus the Middle one, batch 4):

  * every distinct convolution shape of that UNet at batch 4 through the C ABI: images are independent of
    their batch bit for bit (forward, input gradient), parameter gradients add up over the images, and the
    last image is checked against a CPU convolution (fp64; fp32 for the layers at 512x512 and above, where
    fp64 on the host would take minutes);
  * attention at the ffhq sizes (S = 1024 tokens, C = 256) forward and backward against fp64;
  * the ffhq WIRING (13 down modules with attention before the last DownBlock, 13 up modules with attention
    after the first UpBlock, additive skips) at a patch the CPU can evaluate (128x128, attention
    resolutions scaled alike) against the oracle in fp64: prediction, loss, global gradient error;
  * the full 1024x1024 model at batch 4: eval forward independent of the batch, training loss reproducible
    bit for bit;
  * a 1000-step sampling run (what BASELINE's ffhq configuration asks of the sampler) on the small golden
    model against the oracle's sampler, with a stated per-pixel bound - drift over 1000 steps is the risk.
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
from vision_amd.presets import get_preset            # noqa: E402

from conftest import load_golden, sub                # noqa: E402
from test_kernels_gpu import relerr, pack_weights    # noqa: E402
from test_full_size_gpu import _run, _release_temporaries, BF, DT   # noqa: E402,F401  (autouse fixture)

DEV = torch.device("cuda:0")

# B, H, W, Cin, Cout, k, stride, up: the distinct layer shapes of the ffhq UNet
FFHQ_CASES = [
    (4, 1024, 1024, 3, 64, 3, 1, 0), (4, 1024, 1024, 3, 64, 1, 1, 0),
    (4, 1024, 1024, 64, 64, 3, 1, 0), (4, 1024, 1024, 64, 64, 3, 2, 0),
    (4, 512, 512, 64, 64, 3, 1, 0), (4, 512, 512, 64, 64, 3, 2, 0),
    (4, 256, 256, 64, 128, 3, 1, 0), (4, 256, 256, 64, 128, 1, 1, 0),
    (4, 256, 256, 128, 128, 3, 1, 0), (4, 256, 256, 128, 128, 3, 2, 0),
    (4, 128, 128, 128, 128, 3, 1, 0), (4, 128, 128, 128, 128, 3, 2, 0),
    (4, 64, 64, 128, 256, 3, 1, 0), (4, 64, 64, 128, 256, 1, 1, 0),
    (4, 64, 64, 256, 256, 3, 1, 0), (4, 64, 64, 256, 256, 3, 2, 0),
    (4, 32, 32, 256, 256, 3, 1, 0), (4, 32, 32, 256, 256, 1, 1, 0), (4, 32, 32, 256, 256, 3, 2, 0),
    (4, 16, 16, 256, 256, 3, 1, 0), (4, 16, 16, 256, 256, 1, 1, 0),
    (4, 16, 16, 256, 256, 3, 1, 1), (4, 32, 32, 256, 256, 3, 1, 1), (4, 64, 64, 128, 128, 3, 1, 1),
    (4, 128, 128, 128, 128, 3, 1, 1), (4, 256, 256, 64, 64, 3, 1, 1), (4, 512, 512, 64, 64, 3, 1, 1),
    (4, 64, 64, 256, 128, 3, 1, 0), (4, 64, 64, 256, 128, 1, 1, 0), (4, 64, 64, 128, 128, 3, 1, 0),
    (4, 256, 256, 128, 64, 3, 1, 0), (4, 256, 256, 128, 64, 1, 1, 0), (4, 256, 256, 64, 64, 3, 1, 0),
    (4, 1024, 1024, 64, 3, 3, 1, 0), (4, 1024, 1024, 64, 3, 1, 1, 0),
    (4, 1024, 1024, 3, 3, 3, 1, 0), (4, 1024, 1024, 3, 3, 1, 1, 0),
]


@pytest.mark.parametrize("case", FFHQ_CASES, ids=lambda c: "x".join(map(str, c)))
def test_ffhq_conv_layers(case, no_splitk):
    B, H, W, Cin, Cout, k, stride, up = case
    gen = torch.Generator().manual_seed(13)
    pad = 1 if k == 3 else 0
    He, We = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (He + 2 * pad - k) // stride + 1, (We + 2 * pad - k) // stride + 1
    dgen = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn((B, H, W, Cin), device=DEV, generator=dgen).to(BF)
    dy = torch.randn((B, Ho, Wo, Cout), device=DEV, generator=dgen).to(BF)
    w64 = (torch.randn((Cout, Cin, k, k), generator=gen) / math.sqrt(Cin * k * k)).to(BF).double()
    bias = torch.randn(Cout, generator=gen).to(DEV)
    wf, wd = pack_weights(w64, BF, DT)
    y, dx, dw, db = _run(case, B, x, wf, wd, dy, bias)

    dw_sum = torch.zeros_like(dw, dtype=torch.float64)
    db_sum = torch.zeros_like(db, dtype=torch.float64)
    last = None
    for i in range(B):
        sl = slice(i, i + 1)
        last = _run(case, 1, x[sl], wf, wd, dy[sl], bias)
        assert torch.equal(last[0], y[sl]), f"forward of image {i} depends on the batch"
        assert torch.equal(last[1], dx[sl]), f"input gradient of image {i} depends on the batch"
        dw_sum += last[2].double()
        db_sum += last[3].double()
    assert relerr(dw, dw_sum) < 1e-5
    assert relerr(db, db_sum) < 1e-5

    # the last image against a CPU convolution: fp64, or fp32 where fp64 would take minutes on the host
    big = 2.0 * Ho * Wo * Cin * Cout * k * k > 8e9
    rt = torch.float32 if big else torch.float64
    xl = x[-1:].permute(0, 3, 1, 2).to(rt).cpu().requires_grad_(True)
    wl = w64.to(rt).clone().requires_grad_(True)
    xin = F.interpolate(xl, scale_factor=2, mode="nearest") if up else xl
    yl = F.conv2d(xin, wl, bias.to(rt).cpu(), stride=stride, padding=pad)
    dyl = dy[-1:].permute(0, 3, 1, 2).to(rt).cpu()
    gx, gw = torch.autograd.grad(yl, [xl, wl], dyl)
    assert relerr(y[-1:].permute(0, 3, 1, 2), yl.detach()) < 4e-3
    assert relerr(dx[-1:].permute(0, 3, 1, 2), gx) < 4e-3
    assert relerr(last[2], gw) < (1e-3 if big else 2e-4)
    assert relerr(last[3], dyl.double().sum((0, 2, 3))) < 2e-4


@pytest.mark.parametrize("dts", ["f32", "bf16"])
def test_ffhq_attention_block_sizes(dts):
    """AttnBlock at S = 32*32 tokens, C = 256 (reference models/ddpm.py:100-123) forward and backward through
    the engine's attention layer against fp64 autograd of the same formula."""
    from oracle import restatement as R
    cfg = {"name": "ddpm", "input_channels": 3, "learning_rate": 1e-4, "manual_seed": 1, "with_conv": True,
           "sampling_period": 1, "weight_decay": 0.0, "diffusion_steps": 2, "scheduler_gamma": 0.99, "grid": 1,
           "h_dims": [3, 256, 256], "down_index": [0, 1, 0], "attn_res": [32], "patch_size": 64, "dropout": 0.0,
           "latent_dim": [3, 64, 64]}
    # Down [Res 3->256 @64, Down, Res @32, Attn], Middle [Res, Attn, Res], Up [Res @32, Attn, Up, Res 256->3 @64]:
    # three attention blocks at S = 1024, C = 256, one of them on each path like the ffhq model
    tdt = torch.float32 if dts == "f32" else torch.bfloat16
    sd = R.init_state_like_torch(cfg, 1)
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=tdt).to(DEV)
    m.load_state_dict(sd)
    m.train()
    g = torch.Generator().manual_seed(2)
    x, eps, t = torch.rand((2, 3, 64, 64), generator=g), torch.randn((2, 3, 64, 64), generator=g), torch.rand((2, 1, 1, 1), generator=g)
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    ref_loss, ref64 = R.loss_and_grads(sd64, cfg, x.double(), eps.double(), t.double())
    loss = m.p_loss(x.to(DEV), noises=eps.to(DEV), diffusion_times=t.to(DEV))
    loss.backward()
    tol = 1e-5 if dts == "f32" else 2e-2
    assert abs(float(loss.detach()) - float(ref_loss)) < tol * abs(float(ref_loss))
    attn = [n for n, _ in m.named_parameters() if ".q_proj." in n or ".v_proj." in n or ".linear." in n or ".norm." in n]
    assert len(attn) >= 3 * 7
    grads = dict(m.named_parameters())
    want = torch.cat([ref64[n].flatten() for n in attn])
    got = torch.cat([grads[n].grad.detach().flatten().double().cpu() for n in attn])
    err = ((got - want).norm() / want.norm()).item()
    assert err < (2e-3 if dts == "f32" else 6e-2), err


def _ffhq_small(compute_dtype):
    """The ffhq module sequence at 128x128: same widths, levels and attention placement (attention
    resolutions 32 / 16 scaled by the same factor 8 as the patch)."""
    from oracle import restatement as R
    cfg = dict(get_preset("ffhq")["model_config"], patch_size=128, attn_res=[2, 4], latent_dim=[3, 128, 128], dropout=0.0)
    sd = R.init_state_like_torch(cfg, cfg["manual_seed"])
    model = vision_amd.get_model("ddpm", cfg, compute_dtype=compute_dtype).to(DEV)
    model.load_state_dict(sd)
    return model, cfg, sd, R


def test_ffhq_plan_matches_reference_module_sequence():
    from vision_amd.plan import make_plan
    full = make_plan(get_preset("ffhq")["model_config"])
    kinds = lambda p, sec: "".join({"res": "R", "attn": "A", "down": "D", "up": "U"}[l.kind] for l in p.section(sec))
    # module sequences of the unmodified reference on the ffhq YAML
    assert kinds(full, "Downsampling") == "RDRDRDRDRDRAD" and kinds(full, "Middle") == "RAR"
    assert kinds(full, "Upsampling") == "URAURURURURUR"
    small = make_plan(_ffhq_small("fp32")[1])
    for sec in ("Downsampling", "Middle", "Upsampling"):
        assert kinds(small, sec) == kinds(full, sec)
        assert [(l.cin, l.cout) for l in small.section(sec)] == [(l.cin, l.cout) for l in full.section(sec)]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_ffhq_wiring_against_oracle(mode):
    """Tolerances as for the 224x224 model (test_full_size_gpu.test_full_resolution_model_against_oracle):
    prediction 2e-6 / 2e-2, loss 1e-5 / 2e-2, global gradient error 1.5e-3 / 4.5e-2 (fp32 / bf16 compute; twice the measured)."""
    model, cfg, sd, R = _ffhq_small(mode)
    model.train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand((2, 3, 128, 128), generator=g)
    eps = torch.randn((2, 3, 128, 128), generator=g)
    t = torch.rand((2, 1, 1, 1), generator=g)
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    with torch.no_grad():
        pred64 = R.unet_forward({k: v.clone() for k, v in sd64.items()}, cfg, eps.double(), t.double(), training=True)
        bufs = model._ensure_engine().flat_b.clone()
        pred = model(eps.to(DEV), t.to(DEV)).double().cpu()
        model._ensure_engine().flat_b.copy_(bufs)
    assert relerr(pred, pred64) < (2e-6 if mode == "fp32" else 2e-2)
    ref_loss, ref64 = R.loss_and_grads(sd64, cfg, x.double(), eps.double(), t.double())
    loss = model.p_loss(x.to(DEV), noises=eps.to(DEV), diffusion_times=t.to(DEV))
    loss.backward()
    tol = 1e-5 if mode == "fp32" else 2e-2
    assert abs(float(loss.detach()) - float(ref_loss)) < tol * abs(float(ref_loss))
    names = [n for n, _ in model.named_parameters()]
    want = torch.cat([ref64[n].flatten() for n in names])
    got = torch.cat([p.grad.detach().flatten().double().cpu() for _, p in model.named_parameters()])
    err = ((got - want).norm() / want.norm()).item()
    print(f"global gradient error, ffhq wiring at 128x128, {mode}: {err:.2e}")
    assert err < (1.5e-3 if mode == "fp32" else 4.5e-2), err     # measured 7.3e-4 / 2.2e-2
    # eval-mode sampler on the same wiring: 3 steps against the oracle's sampler (from the initial state: the
    # training step above has moved the BatchNorm running statistics)
    model.load_state_dict(sd)
    model.eval()
    noise = torch.randn((2, 3, 128, 128), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        want_img = R.reverse_diffusion({k: v.clone() for k, v in sd.items()}, cfg, noise, 3)
    got_img = model.generate(2, 3, initial_noise=noise.to(DEV))
    assert relerr(got_img, want_img) < (1e-4 if mode == "fp32" else 3e-2)


def test_ffhq_full_size_batch_independence_and_reproducible_step(no_splitk):
    cfg = dict(get_preset("ffhq")["model_config"], dropout=0.0)
    torch.manual_seed(cfg["manual_seed"])
    model = vision_amd.get_model("ddpm", cfg, compute_dtype="bf16").to(DEV)
    P_, B = cfg["patch_size"], 4
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((B, 3, P_, P_), device=DEV, generator=gen)
    t = torch.rand((B, 1, 1, 1), device=DEV, generator=gen)
    model.eval()
    with torch.no_grad():
        full = model(x, t)
        for i in range(B):
            assert torch.equal(model(x[i:i + 1], t[i:i + 1]), full[i:i + 1]), f"image {i} depends on its batch"
    assert torch.isfinite(full).all()
    model.train()
    eng = model._ensure_engine()
    img = torch.rand((B, 3, P_, P_), device=DEV, generator=gen)
    eps = torch.randn((B, 3, P_, P_), device=DEV, generator=gen)
    tt = torch.rand((B, 1, 1, 1), device=DEV, generator=gen)
    bufs = eng.flat_b.clone()
    runs = []
    for _ in range(2):
        eng.flat_b.copy_(bufs)
        eng.flat_g.zero_()
        loss = model.p_loss(img, noises=eps, diffusion_times=tt)
        loss.backward()
        runs.append((loss.detach().clone(), eng.flat_g.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    assert torch.isfinite(runs[0][1]).all()
    err = ((runs[0][1] - runs[1][1]).norm() / runs[0][1].norm()).item()
    assert err < 1e-6, err


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_thousand_step_sampler_against_oracle(mode):
    """T = 1000 denoising steps from fixed noise on the golden t0 model (trained state) against the oracle's
    fp32 sampler.  Per-pixel bound, relative to max|reference|: 5e-5 (fp32 compute: rounding differences of
    1000 chained forwards; measured 1e-5) and 1e-2 (bf16 compute; measured 3.4e-3)."""
    from oracle import restatement as R
    g, cfg, B = load_golden("t0")
    sd = sub(g, "state3")
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=mode).to(DEV)
    m.load_state_dict(sd)
    m.eval()
    noise = g["gen_noise"]
    with torch.no_grad():
        want = R.reverse_diffusion({k: v.clone() for k, v in sd.items()}, cfg, noise.clone(), 1000)
    got = m.generate(noise.shape[0], 1000, initial_noise=noise.to(DEV)).cpu()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f"1000-step sampler, {mode}: max abs error / max|ref| = {err:.3e}")
    assert err < (5e-5 if mode == "fp32" else 1e-2), err
