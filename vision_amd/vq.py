"""Vector-quantisation block with the interface of the reference `VQBlock`
(reference models/vqvaeunet.py:104-146; the same block in models/vqvae.py:13-55), running on libvision_amd.so.

    block = VQBlock(num_embed, embed_dim, beta)
    quantized, vq_loss = block(latents)            # latents [B, D, H, W] fp32 on the GPU

`block.embedding.weight` is the [K, D] codebook under the reference's state_dict key (`embedding.weight`) with
the reference's initialisation (uniform in +-1/K, vqvaeunet.py:113).  There is no CPU path: the lookup, the
straight-through output, the loss and both gradients are HIP kernels (csrc/vq.hip).
"""
import torch
from torch import nn

from ._lib import call, lib


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _VQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latents, weight, beta):
        if not latents.is_cuda or not weight.is_cuda:
            raise RuntimeError("VQBlock runs on the GPU only (libvision_amd.so has no CPU path)")
        if latents.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError("VQBlock computes in fp32: latents and codebook must be float32")
        z = latents.contiguous()
        w = weight.contiguous()
        B, D, H, W = z.shape
        K = w.shape[0]
        if w.shape[1] != D:
            raise ValueError(f"latents have {D} channels, the codebook {w.shape[1]} features")
        M = B * H * W
        st = torch.cuda.current_stream().cuda_stream
        ws = torch.empty(lib.vamd_vq_workspace(M), dtype=torch.uint8, device=z.device)
        idx = torch.empty(M, dtype=torch.int64, device=z.device)
        out = torch.empty_like(z)
        loss = torch.empty(2, dtype=torch.float32, device=z.device)
        call("vamd_vq_forward", z.data_ptr(), w.data_ptr(), M, D, K, H * W, 1, float(beta), idx.data_ptr(), out.data_ptr(),
             loss.data_ptr(), ws.data_ptr(), ws.numel(), st)
        ctx.save_for_backward(z, w, idx)
        ctx.beta = float(beta)
        ctx.mark_non_differentiable(idx)
        return out, loss[0], idx

    @staticmethod
    def backward(ctx, g_out, g_loss, _g_idx):
        z, w, idx = ctx.saved_tensors
        B, D, H, W = z.shape
        K = w.shape[0]
        st = torch.cuda.current_stream().cuda_stream
        need_z, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_out = g_out.contiguous().float() if g_out is not None else None
        g_loss = g_loss.reshape(1).contiguous().float() if g_loss is not None else None
        dz = torch.empty_like(z) if need_z else None
        dw = torch.empty_like(w) if need_w else None
        call("vamd_vq_backward", z.data_ptr(), w.data_ptr(), idx.data_ptr(), _ptr(g_out), _ptr(g_loss), B * H * W, D, K,
             H * W, 1, ctx.beta, _ptr(dz), _ptr(dw), st)
        return dz, dw, None


class VQBlock(nn.Module):
    def __init__(self, num_embed: int, embed_dim: int, beta: float = 0.25):
        super().__init__()
        self.K = num_embed
        self.D = embed_dim
        self.beta = beta
        self.embedding = nn.Embedding(self.K, self.D)
        self.embedding.weight.data.uniform_(-1 / self.K, 1 / self.K)
        self.last_indices = None           # [B*H*W] int64 codes of the latest forward (rows in b, h, w order)

    def forward(self, latents):
        out, loss, idx = _VQFn.apply(latents, self.embedding.weight, self.beta)
        self.last_indices = idx
        return out, loss
