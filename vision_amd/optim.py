"""AdamW over the engine's flat parameter / gradient arenas: one HBM-bound kernel launch per step
(reads p, g, m, v and writes p, m, v once) instead of a per-tensor loop."""
import torch

from ._lib import call
from .engine import _ptr, _stream


class FusedAdamW(torch.optim.Optimizer):
    """Same update rule and defaults as torch.optim.AdamW (reference models/ddpm.py:340-345).
    `param_groups[0]['lr']` is honoured each step, so torch LR schedulers work unchanged."""

    kernel = "vamd_adamw"

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self._model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._m = self._v = None
        self._steps = 0
        self.grad_scale = 1.0            # 1/world_size when gradients were summed, not averaged

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._model._ensure_engine()
        if self._m is None or self._m.data_ptr() == 0 or self._m.numel() != eng.flat_p.numel() \
                or self._m.device != eng.flat_p.device:
            self._m = torch.zeros_like(eng.flat_p)
            self._v = torch.zeros_like(eng.flat_p)
        reducer = getattr(self._model, "grad_reducer", None)
        if reducer is not None:
            reducer.wait()               # gradient all-reduce (side stream) must have landed
        g = self.param_groups[0]
        self._steps += 1
        call(self.kernel, _ptr(eng.flat_p), _ptr(eng.flat_g), _ptr(self._m), _ptr(self._v), eng.flat_p.numel(),
             float(g["lr"]), None, float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
             float(g["weight_decay"]), self._steps, float(self.grad_scale), _stream())
        eng.dirty = True
        return loss

    def zero_grad(self, set_to_none=True):
        # gradients are views of one arena that backward overwrites; keep the views, clear the arena
        eng = self._model._ensure_engine()
        eng.flat_g.zero_()
        eng.attach_grads()


class FusedAdamL2(FusedAdamW):
    """torch.optim.Adam with its L2 weight decay (g + weight_decay * p enters the moments; reference
    models/vqvaeunet.py:263-269), same arena-wide launch."""
    kernel = "vamd_adam_l2"

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
