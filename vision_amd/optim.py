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
        # capturable: the learning rate and the step count live in device memory and the step count is incremented by a
        # launch of its own, so that a step captured into a HIP graph (vision_amd.graphed) updates correctly on replay
        self.capturable = False
        self._lr_dev = self._steps_dev = None
        self._lr_on_dev = None

    def make_capturable(self):
        eng = self._model._ensure_engine()
        self.capturable = True
        self._steps_dev = torch.full((1,), self._steps, dtype=torch.int64, device=eng.device)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=eng.device)
        self.sync_lr()

    def sync_lr(self):
        """Bring the device copy of the learning rate up to date (outside any capture; a scheduler may have moved it)."""
        lr = float(self.param_groups[0]["lr"])
        if self._lr_on_dev != lr:
            self._lr_dev.fill_(lr)
            self._lr_on_dev = lr

    def state_steps(self):
        """Optimiser steps taken so far (read from the device in capturable mode: replays do not pass through step())."""
        return int(self._steps_dev.item()) if self.capturable else self._steps

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
        if self.capturable:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_lr()           # (a captured step's caller does this before every replay)
            self._steps_dev.add_(1)
            call("vamd_adam_dev", _ptr(eng.flat_p), _ptr(eng.flat_g), _ptr(self._m), _ptr(self._v), eng.flat_p.numel(),
                 _ptr(self._lr_dev), _ptr(self._steps_dev), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                 float(g["weight_decay"]), float(self.grad_scale), int(self.kernel == "vamd_adam_l2"), _stream())
        else:
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
