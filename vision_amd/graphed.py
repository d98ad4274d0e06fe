"""One whole training step -- objective, forward, hand-written backward, optimiser, weight re-pack -- captured ONCE
into a HIP graph and replayed: the ~400-600 launches of a step become one call on the host (0.13-0.17 ms instead of
4-12 ms of launch issue), which is what keeps a step device-bound when the host also has collectives to drive.  The
device executes the same kernels in the same order; a replay takes as long as the eager step (measured: flowers 7.08
vs 7.02 ms, latent diffusion 5.58 vs 5.55 ms), so this buys host time, not device time.

What makes a step capturable here: every kernel takes its stream from torch's current stream; the engine allocates
through torch's caching allocator (a capture gets its own pool, so replays see the same addresses); torch's device
generator is graph-aware (noise, diffusion times and dropout masks advance per replay); nothing in the step reads a
value back to the host; and the optimiser keeps its step count and learning rate on the device (FusedAdamW.capturable).
The per-step bookkeeping of `training_step` (reference models/ddpm.py:314-316: `self.log(...)`, `train_losses.append`)
runs on the host after every replay, on a copy of the captured step's outputs.

Not captured: a step with a live gradient exchange over a backend that stages through the host (gloo).  With RCCL the
collectives are stream work and capture like kernels, but that combination has not run on hardware here, so it has to
be asked for (VAMD_GRAPH_DP=1).
"""
import os

import torch


class GraphedTrainStep:
    """stepper = GraphedTrainStep(model, optimizer); loss = stepper(batch, batch_idx) in place of
    `loss = model.training_step(batch, batch_idx); loss.backward(); optimizer.step()`.  The first `warmup` calls run
    eagerly (they are ordinary training steps and initialise what is built lazily), the next call captures and replays."""

    def __init__(self, model, optimizer, warmup=2):
        self.model, self.opt, self.warmup = model, optimizer, warmup
        self.graph = None
        self._calls = 0
        self._static = None            # the captured batch's tensors (inputs are copied into them before a replay)
        self._loss = None
        self._logged = []              # (name, static tensor, kwargs) of every model.log call of the captured step
        self._appended = []            # (list attribute, static tensor) of every tensor the step appended to a list
        red = getattr(model, "grad_reducer", None)
        if red is not None and getattr(red, "_on", False) and os.environ.get("VAMD_GRAPH_DP") != "1":
            raise RuntimeError("GraphedTrainStep: a live gradient exchange is captured only on request (VAMD_GRAPH_DP=1, RCCL)")
        if not optimizer.capturable:
            optimizer.make_capturable()

    def _eager(self, batch, idx):
        loss = self.model.training_step(batch, idx)
        loss.backward()
        self.opt.step()
        return loss

    def _capture(self, batch, idx):
        model, opt = self.model, self.opt
        self._static = tuple(t.clone() if torch.is_tensor(t) else t for t in batch)
        lists = {k: v for k, v in vars(model).items() if isinstance(v, list)}
        before = {k: len(v) for k, v in lists.items()}
        logged = []
        model.log = lambda name, value, **kw: logged.append((name, value, kw))      # shadows the class's method
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self._loss = self._eager(self._static, idx)
        finally:
            del model.log
        self._logged = [(n, v, kw) for n, v, kw in logged if torch.is_tensor(v)]
        for k, v in lists.items():
            self._appended += [(k, t) for t in v[before[k]:] if torch.is_tensor(t)]
            del v[before[k]:]           # a capture executes nothing: these entries hold no step's values
        opt._steps -= 1                  # ... and no optimiser step was taken (the device counter did not move either)

    def __call__(self, batch, idx=0):
        """One training step on `batch` (same shapes on every call); returns this step's loss (a copy of the graph's)."""
        self._calls += 1
        self.opt.sync_lr()
        if self._calls <= self.warmup:
            return self._eager(batch, idx)
        if self.graph is None:
            self._capture(batch, idx)
        for dst, src in zip(self._static, batch):
            if torch.is_tensor(dst):
                if dst.shape != src.shape:
                    raise ValueError(f"GraphedTrainStep was captured for a batch of shape {tuple(dst.shape)}, got {tuple(src.shape)}")
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.opt._steps += 1
        copies = {}

        def copy_of(t):
            if id(t) not in copies:
                copies[id(t)] = t.detach().clone()
            return copies[id(t)]
        for name, t, kw in self._logged:
            self.model.log(name, copy_of(t), **kw)
        for attr, t in self._appended:
            getattr(self.model, attr).append(copy_of(t))
        return copy_of(self._loss)
