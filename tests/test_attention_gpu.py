"""Fused attention core (csrc/attention.hip, reference models/ddpm.py:113-120) against fp64 and against the separate
score / softmax / product kernels it replaces in forwards without a backward."""
import ctypes as C
import math

import pytest
import torch

from vision_amd import _lib
from vision_amd._lib import call, lib

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

SHAPES = [(2, 196, 512), (4, 1024, 256), (3, 49, 64), (1, 1, 32), (2, 17, 96), (1, 1000, 512), (64, 196, 512)]


def _st():
    return torch.cuda.current_stream().cuda_stream


def run_fused(q, k, v):
    B, S, Cc = q.shape
    Sp = (S + 7) // 8 * 8
    vT = torch.full((B, Cc, Sp), float("nan"), dtype=q.dtype, device=DEV)          # padding must never be read
    vT[:, :, :S] = v.transpose(1, 2)
    o = torch.full_like(q, 7.0)
    call("vamd_attention_fwd", q.data_ptr(), k.data_ptr(), vT.data_ptr(), o.data_ptr(), B, S, Cc, Sp, Cc ** -0.5, _lib.BF16, _st())
    return o


def run_separate(q, k, v):
    """vamd_bmm_nt -> vamd_softmax_rows -> vamd_bmm_nt, as the engine's attention layer calls them."""
    B, S, Cc = q.shape
    Sp = (S + 7) // 8 * 8
    scores = torch.empty((B, S, Sp), dtype=torch.float32, device=DEV)
    call("vamd_bmm_nt", q.data_ptr(), k.data_ptr(), scores.data_ptr(), B, S, S, Cc, Cc, Cc, Sp, Cc ** -0.5, _lib.BF16, 1, _lib.ALGO_AUTO, _st())
    P = torch.zeros((B, S, Sp), dtype=q.dtype, device=DEV)
    call("vamd_softmax_rows", scores.data_ptr(), P.data_ptr(), B * S, S, Sp, _lib.BF16, _st())
    vT = torch.zeros((B, Cc, Sp), dtype=q.dtype, device=DEV)
    vT[:, :, :S] = v.transpose(1, 2)
    o = torch.empty_like(q)
    call("vamd_bmm_nt", P.data_ptr(), vT.data_ptr(), o.data_ptr(), B, S, Cc, S, Sp, Sp, Cc, 1.0, _lib.BF16, 0, _lib.ALGO_AUTO, _st())
    return o


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_fused_attention_core(shape):
    B, S, Cc = shape
    g = torch.Generator().manual_seed(sum(shape))
    q = (torch.randn((B, S, Cc), generator=g) * 1.5).to(DEV).bfloat16()
    k = (torch.randn((B, S, Cc), generator=g) * 1.5).to(DEV).bfloat16()
    v = torch.randn((B, S, Cc), generator=g).to(DEV).bfloat16()
    assert lib.vamd_attention_fwd_supported(S, Cc, _lib.BF16) == 1
    o = run_fused(q, k, v)
    assert torch.isfinite(o).all()
    qd, kd, vd = q.double().cpu(), k.double().cpu(), v.double().cpu()
    want = torch.softmax(qd @ kd.transpose(1, 2) * Cc ** -0.5, dim=-1) @ vd
    err = ((o.double().cpu() - want).norm() / want.norm()).item()
    assert err < 6e-3, err                       # bf16 probabilities and outputs
    sep = run_separate(q, k, v)
    err_sep = ((sep.double().cpu() - want).norm() / want.norm()).item()
    # same MFMA accumulation order over channels / keys and the row-softmax kernel's own arithmetic: the same bits
    assert torch.equal(o, sep), (err, err_sep)


def test_fused_attention_limits():
    assert lib.vamd_attention_fwd_supported(1025, 256, _lib.BF16) == 0
    assert lib.vamd_attention_fwd_supported(196, 520, _lib.BF16) == 0
    assert lib.vamd_attention_fwd_supported(196, 512, _lib.F32) == 0
    z = torch.zeros(64, device=DEV)
    rc = lib.vamd_attention_fwd(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 2000, 64, 2000, C.c_float(1.0), _lib.BF16, None)
    assert rc != 0 and b"not covered" in lib.vamd_last_error()


@pytest.mark.parametrize("name", ["t0", "t1"])
def test_engine_eval_forward_with_fused_attention(name, monkeypatch):
    """The whole eval-mode forward with the attention core fused against the same forward on the separate kernels."""
    import vision_amd
    from vision_amd import engine as E
    from conftest import load_golden, sub
    g, cfg, B = load_golden(name)
    m = vision_amd.get_model("ddpm", cfg, compute_dtype=torch.bfloat16)
    m.load_state_dict(sub(g, "state3"))
    m = m.to(DEV).eval()
    out, calls = {}, []
    orig = E.call
    monkeypatch.setattr(E, "call", lambda n, *a: (calls.append(n), orig(n, *a))[1])
    for fuse in (True, False):
        monkeypatch.setattr(E, "FUSE_ATTENTION", fuse)
        calls.clear()
        with torch.no_grad():
            out[fuse] = m(g["x_t"].to(DEV), (g["sched_n"] ** 2).to(DEV)).clone()
        assert ("vamd_attention_fwd" in calls) == fuse and ("vamd_softmax_rows" in calls) != fuse
    assert torch.equal(out[True], out[False])
