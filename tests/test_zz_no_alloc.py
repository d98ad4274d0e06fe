"""vamd_gn_coef_noalloc (GroupNorm coefficients with the partial sums in a buffer built into the library)
against the two-call form with a caller-allocated workspace (vamd_chan_stats + vamd_gn_finalize), through
the C ABI: bit-identical scale / shift / mean / rstd on two calls in a row, the second with a smaller
input (the built-in buffer then still holds the first call's sums), plus the size limit's error."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from vision_amd import _lib                # noqa: E402
from vision_amd._lib import call, lib      # noqa: E402

DEV = torch.device("cuda:0")
P = lambda t: None if t is None else t.data_ptr()
S = lambda: torch.cuda.current_stream().cuda_stream


def _both(B, HW, Cc, tdt, dt, seed):
    gen = torch.Generator().manual_seed(seed)
    x = (torch.randn((B, HW, Cc), generator=gen) * 1.5 + 0.3).to(tdt).to(DEV)
    gamma = (torch.rand(Cc, generator=gen) + 0.5).to(DEV)
    beta = torch.randn(Cc, generator=gen).to(DEV)
    nchunk = lib.vamd_stats_chunks(HW, B)
    part = torch.empty((B, nchunk, Cc, 2), dtype=torch.float32, device=DEV)
    want = torch.empty((4, B, Cc), dtype=torch.float32, device=DEV)
    call("vamd_chan_stats", P(x), P(part), B, HW, Cc, dt, S())
    call("vamd_gn_finalize", P(part), nchunk, B, Cc, 32, HW, P(gamma), P(beta), 1e-6, P(want[0]), P(want[1]), P(want[2]),
         P(want[3]), S())
    got = torch.full((4, B, Cc), float("nan"), dtype=torch.float32, device=DEV)
    call("vamd_gn_coef_noalloc", P(x), B, Cc, 32, HW, P(gamma), P(beta), 1e-6, P(got[0]), P(got[1]), P(got[2]), P(got[3]),
         dt, S())
    torch.cuda.synchronize()
    return got, want, x


@pytest.mark.parametrize("dts", ["bf16", "f32"])
def test_no_alloc_groupnorm_matches_workspace_form_twice(dts):
    tdt, dt = {"bf16": (torch.bfloat16, _lib.BF16), "f32": (torch.float32, _lib.F32)}[dts]
    # first call: the attention input of the 1024x1024 model (32x32 tokens, 256 channels), 4 images
    got, want, x = _both(4, 32 * 32, 256, tdt, dt, 1)
    assert torch.equal(got, want)
    # the coefficients are GroupNorm's: check one against torch on the host
    ref_mean = x.double().cpu().reshape(4, 1024, 32, 8).mean(dim=(1, 3))
    assert (got[2].cpu().double().reshape(4, 32, 8)[:, :, 0] - ref_mean).abs().max() < 1e-4
    # second call, smaller input (fewer images, rows, channels and chunks)
    got2, want2, _ = _both(2, 14 * 14, 64, tdt, dt, 2)
    assert torch.equal(got2, want2)
    # and a larger one again
    got3, want3, _ = _both(3, 64 * 64, 512, tdt, dt, 3)
    assert torch.equal(got3, want3)


def test_no_alloc_groupnorm_rejects_shapes_beyond_the_built_in_buffer():
    B, HW, Cc = 8, 256 * 256, 1024                      # 8 * 256 chunks * 1024 channels * 2 floats = 16 MiB > 8 MiB
    x = torch.zeros((B, 8, Cc), dtype=torch.bfloat16, device=DEV)      # never read: the size check comes first
    g = torch.ones(Cc, device=DEV)
    out = torch.empty((4, B, Cc), dtype=torch.float32, device=DEV)
    rc = lib.vamd_gn_coef_noalloc(P(x), B, Cc, 32, HW, P(g), P(g), 1e-6, P(out[0]), P(out[1]), P(out[2]), P(out[3]),
                                  _lib.BF16, S())
    assert rc == -1
    msg = lib.vamd_last_error().decode()
    assert "built-in" in msg and "vamd_chan_stats" in msg
