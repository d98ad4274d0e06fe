"""TEST INFRASTRUCTURE ONLY -- regenerates tests/golden/vq.safetensors.

Runs the UNMODIFIED reference `VQBlock` (models/vqvaeunet.py:104-146, loaded by oracle/ref_loader.py) on three
kinds of input and records inputs, code indices (torch.argmin is observed while the block runs), outputs, loss and
gradients.  Only runs where the reference is mounted (the build container):

    python -m oracle.make_golden_vq

grid/     values on a dyadic grid, so every fp32 sum of the distance is exact whatever its order: duplicated code
          rows and latents half-way between two codes are EXACT ties, and the recorded index is the first minimum.
trained/  a codebook of unit scale and latents near their codes (best two distances far apart): indices cannot
          depend on summation order.  Gradients of sum(out * wgt) + 3 * vq_loss recorded.
init/     the reference's own initialisation (uniform +-1/K, vqvaeunet.py:113) with unit-variance latents: the
          best distances differ by less than fp32 rounding of |z|^2, so the recorded indices depend on torch's
          matmul blocking; tests compare these rows by distance, not by index.
"""
import os

import torch
from safetensors.torch import save_file

from oracle.ref_loader import load_reference_vqblock

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vq.safetensors")


class _ArgminRecorder:
    def __enter__(self):
        self.seen = []
        self._orig = torch.argmin

        def patched(*a, **k):
            r = self._orig(*a, **k)
            self.seen.append(r.detach().clone())
            return r

        torch.argmin = patched
        return self

    def __exit__(self, *exc):
        torch.argmin = self._orig


def run(block, lat, wgt=None):
    lat = lat.clone().requires_grad_(True)
    with _ArgminRecorder() as rec:
        out, loss = block(lat)
    res = {"indices": rec.seen[0].view(-1), "out": out.detach(), "vq_loss": loss.detach().reshape(1)}
    if wgt is not None:
        block.zero_grad()
        ((out * wgt).sum() + 3.0 * loss).backward()
        res["grad_latents"] = lat.grad.clone()
        res["grad_weight"] = block.embedding.weight.grad.clone()
    return res


def main():
    ref = load_reference_vqblock()
    out = {}
    g = torch.Generator().manual_seed(5)

    # grid: multiples of 1/8 in [-1, 1]; D = 64
    K, D, B, H = 96, 64, 2, 5
    E = torch.randint(-8, 9, (K, D), generator=g).float() / 8
    E[40] = E[10]
    E[4] = E[3]
    E[77] = E[3]
    z = torch.randint(-8, 9, (B * H * H, D), generator=g).float() / 8
    z[0] = E[40]                       # distance 0 to codes 10 and 40
    z[1] = E[77]                       # ... to 3, 4 and 77
    z[2] = (E[20] + E[50]) / 2         # half-way between two codes
    z[3] = (E[60] + E[5]) / 2
    z[7] = E[95]
    z[8] = E[0]
    blk = ref.VQBlock(K, D)
    blk.embedding.weight.data.copy_(E)
    lat = z.view(B, H, H, D).permute(0, 3, 1, 2).contiguous()
    r = run(blk, lat)
    d64 = ((z.double()[:, None, :] - E.double()[None]) ** 2).sum(-1)
    assert torch.equal(r["indices"], d64.argmin(1)), "grid case is not exact in fp32"
    assert r["indices"][0] == 10 and r["indices"][1] == 3
    out.update({"grid/weight": E, "grid/latents": lat, **{"grid/" + k: v for k, v in r.items()}})

    # trained: unit-scale codebook, latents near their codes
    K, D, B, H = 512, 256, 4, 7
    E = torch.randn((K, D), generator=g) * 0.5
    pick = torch.randint(0, K, (B * H * H,), generator=g)
    z = E[pick] + 0.1 * torch.randn((B * H * H, D), generator=g)
    blk = ref.VQBlock(K, D)
    blk.embedding.weight.data.copy_(E)
    lat = z.view(B, H, H, D).permute(0, 3, 1, 2).contiguous()
    wgt = torch.randn(lat.shape, generator=g)
    r = run(blk, lat, wgt)
    d64 = ((z.double()[:, None, :] - E.double()[None]) ** 2).sum(-1)
    two = d64.topk(2, dim=1, largest=False).values
    assert torch.equal(r["indices"], pick) and float((two[:, 1] - two[:, 0]).min()) > 1.0
    out.update({"trained/weight": E, "trained/latents": lat, "trained/wgt": wgt, **{"trained/" + k: v for k, v in r.items()}})

    # init: the reference's own initial codebook at the butterfly_vqvaeunet sizes
    K, D, B, H = 2048, 256, 2, 7
    torch.manual_seed(42)
    blk = ref.VQBlock(K, D)
    lat = torch.randn((B, D, H, H), generator=g)
    r = run(blk, lat)
    out.update({"init/weight": blk.embedding.weight.detach().clone(), "init/latents": lat,
                **{"init/" + k: v for k, v in r.items()}})

    out = {k: v.contiguous() for k, v in out.items()}
    save_file(out, OUT, metadata={"beta": "0.25", "loss_of_gradients": "sum(out * wgt) + 3 * vq_loss"})
    print(f"{len(out)} tensors, {os.path.getsize(OUT) / 1e6:.2f} MB -> {OUT}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
