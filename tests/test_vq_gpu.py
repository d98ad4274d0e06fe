"""Codebook lookup kernels (csrc/vq.hip through the C ABI and vision_amd.vq.VQBlock) against the reference's
fixtures (tests/golden/vq.safetensors, reference models/vqvaeunet.py:104-146) and the C oracle.  Indices and the
straight-through output are compared bit for bit; the loss and the gradients (fp32 sums whose order torch does
not state) to 2e-6 / 1e-5 relative."""
import ctypes as C
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vq.safetensors")
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def vq():
    from oracle import vq as V
    V.build()
    return V


@pytest.fixture(scope="module")
def golden():
    return load_file(GOLDEN)


def hip_forward(lat, w, nchw=True, beta=0.25):
    """lat [B, D, H, W] (CPU) through vamd_vq_forward in either layout -> (indices, out as [B, D, H, W], loss[2]) on CPU."""
    from vision_amd._lib import call, lib
    B, D, H, W = lat.shape
    M, K = B * H * W, w.shape[0]
    z = (lat if nchw else lat.permute(0, 2, 3, 1)).contiguous().to(DEV)
    wd = w.contiguous().to(DEV)
    ws = torch.empty(lib.vamd_vq_workspace(M), dtype=torch.uint8, device=DEV)
    idx = torch.full((M,), -1, dtype=torch.int64, device=DEV)
    out = torch.full_like(z, 7.0)
    loss = torch.full((2,), 7.0, device=DEV)
    call("vamd_vq_forward", z.data_ptr(), wd.data_ptr(), M, D, K, H * W, int(nchw), beta, idx.data_ptr(), out.data_ptr(),
         loss.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    out = out.cpu()
    return idx.cpu(), (out if nchw else out.permute(0, 3, 1, 2).contiguous()), loss.cpu()


@pytest.mark.parametrize("nchw", [True, False], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("case", ["grid", "trained"])
def test_indices_bit_exact_against_reference(golden, case, nchw):
    lat, w = golden[case + "/latents"], golden[case + "/weight"]
    idx, out, loss = hip_forward(lat, w, nchw)
    assert torch.equal(idx, golden[case + "/indices"])
    assert torch.equal(out, golden[case + "/out"])
    ref = float(golden[case + "/vq_loss"])
    assert abs(float(loss[0]) - ref) <= 2e-6 * abs(ref)
    assert abs(float(loss[1]) * 1.25 - ref) <= 2e-6 * abs(ref)


def test_near_ties_match_the_stated_order(vq, golden):
    """Reference initialisation: distances closer than fp32 rounding.  The kernel evaluates the stated order, so it
    equals the C oracle bit for bit; against the reference's torch indices the chosen codes are compared by exact
    distance (both within 8 ulp(256) of the minimum)."""
    lat, w = golden["init/latents"], golden["init/weight"]
    idx, out, loss = hip_forward(lat, w)
    oidx, oout, oloss, _ = vq.forward(lat, w)
    assert torch.equal(idx, oidx) and torch.equal(out, oout)
    flat = lat.double().permute(0, 2, 3, 1).reshape(-1, lat.shape[1])
    d = (flat * flat).sum(1, keepdim=True) + (w.double() ** 2).sum(1) - 2 * flat @ w.double().t()
    assert float((d.gather(1, idx[:, None])[:, 0] - d.min(1).values).max()) < 8 * 2.0 ** -15
    assert abs(float(loss[0]) - float(golden["init/vq_loss"])) <= 1e-5 * abs(float(golden["init/vq_loss"]))
    agree = float((idx == golden["init/indices"]).float().mean())
    print(f"near-tie rows choosing the same code as torch's CPU matmul: {agree:.2%}")


@pytest.mark.parametrize("shape", [(16, 256, 7, 7, 2048), (3, 70, 5, 4, 300), (1, 8, 1, 1, 5), (2, 33, 9, 9, 129)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("nchw", [True, False], ids=["nchw", "nhwc"])
def test_against_oracle_at_config_and_ragged_sizes(vq, shape, nchw):
    """butterfly_vqvaeunet sizes (batch 16, latent 256 x 7 x 7, 2048 codes) and sizes that fill no tile; latents of
    codebook scale so that distances are close; one NaN row and exact duplicates among the codes."""
    B, D, H, W, K = shape
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn((K, D), generator=g) * 0.02
    lat = torch.randn((B, D, H, W), generator=g) * 0.02
    if K > 4:
        w[K - 1] = w[1]
        lat[0, :, 0, 0] = w[K - 1]
    if B * H * W > 2:
        lat[-1, 0, -1, -1] = float("nan")
    idx, out, loss = hip_forward(lat, w, nchw)
    oidx, oout, oloss, _ = vq.forward(lat, w)
    assert torch.equal(idx, oidx)
    assert torch.equal(out.nan_to_num(nan=123.0), oout.nan_to_num(nan=123.0))
    if K > 4:
        assert idx[0] == 1                                   # the first of the two equal codes
    if B * H * W > 2:
        assert idx[-1] == 0 and torch.isnan(loss[0])         # all distances NaN: the first code, as torch.argmin
    else:
        assert abs(float(loss[0]) - float(oloss[0])) <= 2e-6 * abs(float(oloss[0]))


def test_module_gradients_against_reference(golden):
    from vision_amd.vq import VQBlock
    lat, w, wgt = golden["trained/latents"], golden["trained/weight"], golden["trained/wgt"]
    blk = VQBlock(w.shape[0], w.shape[1], 0.25).to(DEV)
    assert list(blk.state_dict()) == ["embedding.weight"]
    blk.load_state_dict({"embedding.weight": w})
    grads = []
    for _ in range(2):
        blk.zero_grad()
        x = lat.to(DEV).requires_grad_(True)
        out, loss = blk(x)
        ((out * wgt.to(DEV)).sum() + 3.0 * loss).backward()
        grads.append((x.grad.cpu(), blk.embedding.weight.grad.cpu().clone()))
    assert torch.equal(blk.last_indices.cpu(), golden["trained/indices"])
    assert torch.equal(out.detach().cpu(), golden["trained/out"])
    assert torch.allclose(grads[0][0], golden["trained/grad_latents"], rtol=1e-5, atol=1e-9)
    assert torch.allclose(grads[0][1], golden["trained/grad_weight"], rtol=1e-5, atol=1e-9)
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])     # bit-reproducible


def test_reference_initialisation_and_loss_only_gradient():
    """uniform(-1/K, 1/K) codebook (vqvaeunet.py:113); with only vq_loss in the objective the latents receive the
    commitment gradient beta * 2 (z - q) / N and the used codes 2 (q - z) / N summed over their rows."""
    from vision_amd.vq import VQBlock
    torch.manual_seed(0)
    blk = VQBlock(64, 32).to(DEV)
    assert float(blk.embedding.weight.detach().abs().max()) <= 1 / 64
    x = (torch.randn((2, 32, 3, 3), device=DEV) * 0.01).requires_grad_(True)
    out, loss = blk(x)
    loss.backward()
    idx = blk.last_indices
    z = x.detach().permute(0, 2, 3, 1).reshape(-1, 32).double()
    q = blk.embedding.weight.detach().double()[idx]
    n = z.numel()
    want_z = (0.25 * 2 * (z - q) / n).view(2, 3, 3, 32).permute(0, 3, 1, 2)
    want_w = torch.zeros(64, 32, dtype=torch.float64, device=DEV).index_add_(0, idx, 2 * (q - z) / n)
    assert torch.allclose(x.grad.double(), want_z, rtol=1e-5, atol=1e-12)
    assert torch.allclose(blk.embedding.weight.grad.double(), want_w, rtol=1e-5, atol=1e-12)


def test_full_batch_codes_map_to_themselves():
    """Batch 64 of 7 x 7 latents (3136 rows) taken from the codebook itself: every row returns its own code and a
    zero loss, whatever the size."""
    K, D = 2048, 256
    g = torch.Generator().manual_seed(3)
    w = torch.randn((K, D), generator=g)
    pick = torch.randint(0, K, (64 * 49,), generator=g)
    lat = w[pick].view(64, 7, 7, D).permute(0, 3, 1, 2).contiguous()
    idx, out, loss = hip_forward(lat, w)
    assert torch.equal(idx, pick) and torch.equal(out, lat) and float(loss[0]) == 0.0


def test_bad_arguments_are_refused():
    from vision_amd._lib import lib
    z = torch.zeros(8, device=DEV)
    rc = lib.vamd_vq_forward(z.data_ptr(), z.data_ptr(), 10, 4, 2, 3, 1, C.c_float(0.25), z.data_ptr(), z.data_ptr(),
                             z.data_ptr(), z.data_ptr(), 1 << 20, None)
    assert rc != 0 and b"HW" in lib.vamd_last_error()
    rc = lib.vamd_vq_forward(z.data_ptr(), z.data_ptr(), 9, 4, 2, 3, 1, C.c_float(0.25), z.data_ptr(), z.data_ptr(),
                             z.data_ptr(), z.data_ptr(), 8, None)
    assert rc != 0 and b"workspace" in lib.vamd_last_error()
