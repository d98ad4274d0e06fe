"""TEST INFRASTRUCTURE ONLY -- CPU restatement of reference models/vqvaeunet.py:104-146 (VQBlock).

`forward` calls the C restatement (oracle/vq_oracle.c, built by `build()` into oracle/_build/) on the flattened
channels-last rows the reference forms at vqvaeunet.py:116-118; `VQBlockRef` is the same block in torch ops with a
chosen dtype, for gradients (vqvaeunet.py:128-146 line by line).  Only tests, smoke() and the bench's CPU leg may
import this module.
"""
import ctypes as C
import os
import subprocess

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "vq_oracle.c")
LIB = os.path.join(_HERE, "_build", "libvq_oracle.so")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"])
    return LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError(f"{LIB} not built: run __graft_entry__.build() (gcc) first")
        _lib = C.CDLL(LIB)
        _lib.vq_oracle_forward.restype = None
        _lib.vq_oracle_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def forward(latents, weight, beta=0.25):
    """latents [B, D, H, W] fp32, weight [K, D] fp32 (CPU) -> (indices [B*H*W] int64, straight-through output
    [B, D, H, W], loss tensor [2] = (vq_loss, mse), best distance per row [B*H*W])."""
    lat = latents.detach().float().cpu()
    w = weight.detach().float().cpu().contiguous()
    B, D, H, W = lat.shape
    flat = lat.permute(0, 2, 3, 1).contiguous().view(-1, D)            # vqvaeunet.py:116-118
    M, K = flat.shape[0], w.shape[0]
    idx = torch.empty(M, dtype=torch.int64)
    out = torch.empty_like(flat)
    dist = torch.empty(M, dtype=torch.float32)
    loss = torch.empty(2, dtype=torch.float32)
    _load().vq_oracle_forward(flat.data_ptr(), w.data_ptr(), M, D, K, float(beta), idx.data_ptr(), out.data_ptr(),
                              dist.data_ptr(), loss.data_ptr())
    return idx, out.view(B, H, W, D).permute(0, 3, 1, 2).contiguous(), loss, dist      # vqvaeunet.py:146


class VQBlockRef(torch.nn.Module):
    """vqvaeunet.py:104-146 in torch ops of one dtype, taking the code indices as given (so that gradients can be
    compared for a fixed assignment) or computing them the way the reference writes them."""

    def __init__(self, weight, beta=0.25, dtype=torch.float64):
        super().__init__()
        self.weight = torch.nn.Parameter(weight.detach().to(dtype).clone())
        self.beta = beta

    def forward(self, latents, indices=None):
        latents = latents.permute(0, 2, 3, 1).contiguous()
        shape = latents.shape
        flat = latents.view(-1, shape[-1])
        if indices is None:
            dist = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(self.weight ** 2, dim=1) \
                - 2 * torch.matmul(flat, self.weight.t())
            indices = torch.argmin(dist, dim=1)
        one_hot = torch.zeros(indices.numel(), self.weight.shape[0], dtype=self.weight.dtype)
        one_hot.scatter_(1, indices.view(-1, 1), 1)
        q = torch.matmul(one_hot, self.weight).view(shape)
        commitment = F.mse_loss(q.detach(), latents)
        embedding = F.mse_loss(q, latents.detach())
        vq_loss = commitment * self.beta + embedding
        q = latents + (q - latents).detach()
        return q.permute(0, 3, 1, 2).contiguous(), vq_loss
