"""The codebook-lookup oracle (oracle/vq_oracle.c through oracle/vq.py) against the fixtures the reference VQBlock
produced (tests/golden/vq.safetensors, written by oracle/make_golden_vq.py from reference
models/vqvaeunet.py:104-146).  CPU only."""
import os

import pytest
import torch
from safetensors.torch import load_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vq.safetensors")


@pytest.fixture(scope="module")
def vq():
    from oracle import vq as V
    V.build()
    return V


@pytest.fixture(scope="module")
def golden():
    return load_file(GOLDEN)


def exact_distances(lat, w):
    flat = lat.double().permute(0, 2, 3, 1).reshape(-1, lat.shape[1])
    return (flat * flat).sum(1, keepdim=True) + (w.double() ** 2).sum(1) - 2 * flat @ w.double().t()


@pytest.mark.parametrize("case", ["grid", "trained"])
def test_oracle_indices_bit_exact(vq, golden, case):
    """Exact-arithmetic inputs (ties included) and well-separated inputs: same indices, same output bits."""
    lat, w = golden[case + "/latents"], golden[case + "/weight"]
    idx, out, loss, _ = vq.forward(lat, w, 0.25)
    assert torch.equal(idx, golden[case + "/indices"])
    assert torch.equal(out, golden[case + "/out"])
    assert abs(float(loss[0]) - float(golden[case + "/vq_loss"])) <= 2e-6 * abs(float(golden[case + "/vq_loss"]))


def test_oracle_near_ties_by_distance(vq, golden):
    """The reference's initial codebook: neighbouring distances differ by less than fp32 rounding of |z|^2 (~256), so
    the chosen code may differ from torch's blocked matmul; whichever code is chosen must be a minimum up to that
    rounding: within 8 ulp(256) = 2.4e-4 of the exact minimum, as torch's own choice is."""
    lat, w = golden["init/latents"], golden["init/weight"]
    idx, out, loss, _ = vq.forward(lat, w, 0.25)
    d = exact_distances(lat, w)
    best = d.min(1).values
    bound = 8 * 2.0 ** -15
    assert float((d.gather(1, idx[:, None])[:, 0] - best).max()) < bound
    assert float((d.gather(1, golden["init/indices"][:, None])[:, 0] - best).max()) < bound
    assert abs(float(loss[0]) - float(golden["init/vq_loss"])) <= 1e-5 * abs(float(golden["init/vq_loss"]))


def test_reference_gradients_follow_the_stated_formulas(vq, golden):
    """fp64 restatement of the block (oracle.vq.VQBlockRef) against the reference's recorded gradients."""
    lat, w, wgt = golden["trained/latents"], golden["trained/weight"], golden["trained/wgt"]
    blk = vq.VQBlockRef(w, 0.25, torch.float64)
    x = lat.double().requires_grad_(True)
    out, loss = blk(x)
    ((out * wgt.double()).sum() + 3.0 * loss).backward()
    assert torch.allclose(out.float(), golden["trained/out"], rtol=0, atol=1e-6)
    assert torch.allclose(x.grad.float(), golden["trained/grad_latents"], rtol=1e-5, atol=1e-9)
    assert torch.allclose(blk.weight.grad.float(), golden["trained/grad_weight"], rtol=1e-5, atol=1e-9)
