"""Data-parallel training over one process per GPU (torch.distributed; backend "nccl" is RCCL on
ROCm, over xGMI inside a node).

What the reference gets implicitly from Lightning's DDP strategy (reference run.py:81-91,
configs/ffhq_ddpm.yaml:27) is made explicit here:
  * per-rank batches and per-rank BatchNorm statistics (no SyncBatchNorm in the reference);
  * gradient MEAN all-reduce, bucketed and launched while backward is still running;
  * BatchNorm buffers broadcast from rank 0 before each training forward (DDP's
    broadcast_buffers=True default);
  * initial parameter broadcast from rank 0.
Sampling shards rows of the initial noise across ranks with no communication.

The engine keeps every gradient in ONE flat fp32 arena laid out in parameter-creation order, and
backward finalises gradients from the arena's tail towards its head (End conv first, the time-MLP
last).  A bucket is therefore a contiguous slice [lo, hi) of the arena: as soon as backward has
passed the layer whose parameters start at `lo`, the slice is complete and its all-reduce is
issued asynchronously; the optimiser waits for all of them.  On a fully connected 8-GPU xGMI mesh
a 25 MB bucket moves in ~0.2 ms, so the exchange hides behind the remaining backward kernels.
"""
import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist


def _staged(t: torch.Tensor, group) -> bool:
    """Device tensors go through the collective directly only with nccl (RCCL).  Any other backend
    (gloo: tests, rehearsals of several ranks on one GPU) gets them staged through host memory."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def broadcast(t: torch.Tensor, src: int, group=None):
    if not _staged(t, group):
        dist.broadcast(t, src=src, group=group)
        return
    host = t.cpu()
    dist.broadcast(host, src=src, group=group)
    t.copy_(host)


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def collectives_on(world: int) -> bool:
    """Collectives are skipped on a single rank, unless VAMD_DP_FORCE=1 asks for them anyway: a
    one-rank process group still goes through the backend (RCCL communicator, streams, async
    handles), which is how the exchange code is rehearsed on a one-GPU machine."""
    if world > 1:
        return True
    return os.environ.get("VAMD_DP_FORCE", "0") == "1" and dist.is_available() and dist.is_initialized()


class GradReducer:
    """Bucketed, overlapped mean all-reduce of a flat gradient arena.

    `boundaries` are arena offsets (ascending) at which a layer's parameters start; gradients
    become final from the last boundary backwards.  GPU tensors go to nccl/RCCL directly; with
    any other backend (gloo) they are staged through host memory, CPU tensors are used as they are."""

    def __init__(self, flat_grads: torch.Tensor, boundaries: List[int], group=None, bucket_bytes=25 << 20,
                 head_bytes=4 << 20):
        self.flat, self.group = flat_grads, group
        self.rank, self.world = world_info(group)
        self.total = flat_grads.numel()
        bset = sorted(set([0] + [b for b in boundaries if 0 < b < self.total]))
        # merge layer boundaries (from the tail) into buckets of at least bucket_bytes
        cut, hi, min_elems = [], self.total, max(1, bucket_bytes // flat_grads.element_size())
        for b in reversed(bset):
            if hi - b >= min_elems or b == 0:
                cut.append((b, hi))
                hi = b
        # The bucket that starts at 0 is complete only when backward ends, so its all-reduce overlaps with
        # nothing: keep it small (the first layers' parameters are few) by splitting it at the highest
        # layer boundary within head_bytes; the part above goes out while the first layers' backward runs.
        head_elems = max(1, head_bytes // flat_grads.element_size())
        if cut and cut[-1][0] == 0 and cut[-1][1] > head_elems:
            inner = [b for b in bset if 0 < b <= head_elems and b < cut[-1][1]]
            if inner:
                lo, hi = cut.pop()
                cut += [(inner[-1], hi), (lo, inner[-1])]
        self.buckets = cut                    # tail first, last one starts at 0
        self._next = 0
        self._pending = []
        self._on = collectives_on(self.world)
        backend = dist.get_backend(group) if self._on else "none"
        self._use_avg = backend == "nccl"     # ReduceOp.AVG exists for nccl/RCCL only
        # VAMD_DP_SERIAL=1: blocking collectives (no overlap with backward), for debugging
        self._serial = os.environ.get("VAMD_DP_SERIAL", "0") == "1"
        # a list here switches on the measurement of the EXPOSED part of the exchange: per step, how long the compute
        # stream has to wait between backward's last kernel and the landing of the last bucket (everything earlier runs
        # beside backward).  Device events on the RCCL path; with staged (host) transport the host time of wait().
        self.timing: Optional[list] = None
        self._t_end = None

    def begin(self):
        self._next = 0
        self._pending = []
        self._t_end = None

    def exposed_ms(self):
        """Per measured step, the milliseconds described at `timing` (call after a device synchronisation)."""
        out = []
        for a, b in self.timing or []:
            out.append(a.elapsed_time(b) if b is not None else a)
        return out

    def layer_done(self, low_offset: int):
        """Gradients at arena offsets >= low_offset are final: issue every bucket that is complete."""
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= low_offset:
            self._issue(*self.buckets[self._next])
            self._next += 1

    def finish(self):
        self.layer_done(0)
        if self.timing is not None and self._on and self.flat.is_cuda and not _staged(self.flat, self.group):
            self._t_end = torch.cuda.Event(enable_timing=True)
            self._t_end.record()             # after backward's last kernel, on the compute stream

    def _issue(self, lo, hi):
        if not self._on:
            return
        view = self.flat[lo:hi]
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        if _staged(view, self.group):
            # fallback transport: the slice is copied to the host (this waits for the kernels that
            # produced it), reduced there, and copied back in wait()
            host = view.cpu()
            work = None if self._serial else dist.all_reduce(host, op=op, group=self.group, async_op=True)
            if self._serial:
                dist.all_reduce(host, op=op, group=self.group)
            self._pending.append((work, view, host))
            return
        # async_op: the collective runs on the backend's own stream after everything already
        # queued on the current stream (the kernels that produced this slice), while later
        # backward kernels keep the compute stream busy
        if self._serial:
            dist.all_reduce(view, op=op, group=self.group)
            self._pending.append((None, view, None))
            return
        work = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        self._pending.append((work, view, None))

    def wait(self):
        """Block the current stream (not the host, on GPU) until every issued bucket has landed."""
        if self._next < len(self.buckets):
            self.finish()
        t_host = time.perf_counter() if (self.timing is not None and self._on and self._t_end is None) else None
        for work, view, host in self._pending:
            if work is not None:
                work.wait()
            if host is not None:
                view.copy_(host)
            if not self._use_avg:
                view.mul_(1.0 / self.world)
        if self._t_end is not None:
            landed = torch.cuda.Event(enable_timing=True)
            landed.record()                  # the compute stream reaches it once every bucket has landed
            self.timing.append((self._t_end, landed))
            self._t_end = None
        elif t_host is not None and self._pending:
            self.timing.append((1e3 * (time.perf_counter() - t_host), None))
        self._pending = []


class DataParallel:
    """Wires a DDPMModel's engine to a GradReducer and mirrors DDP's parameter / buffer broadcasts."""

    def __init__(self, model, group=None, bucket_bytes=25 << 20):
        self.model, self.group = model, group
        self.rank, self.world = world_info(group)
        eng = model._ensure_engine()
        self.engine = eng
        self.reducer = GradReducer(eng.flat_g, eng.layer_offsets(), group, bucket_bytes)
        model.grad_reducer = self.reducer
        eng.reducer = self.reducer
        self._on = collectives_on(self.world)
        if self._on:
            broadcast(eng.flat_p, self._src(), group)       # identical start on every rank
            eng.dirty = True
            self.broadcast_buffers()

    def _src(self):
        return dist.get_global_rank(self.group, 0) if self.group is not None else 0

    def broadcast_buffers(self):
        """Rank 0's BatchNorm running statistics and counters replace everybody's (two collectives:
        the buffers live in one fp32 and one int64 arena)."""
        if not self._on:
            return
        eng = self.engine
        broadcast(eng.flat_b, self._src(), self.group)
        broadcast(eng.flat_nbt, self._src(), self.group)

    def all_reduce_scalar_mean(self, value: torch.Tensor) -> torch.Tensor:
        """`self.log(..., sync_dist=True)` of the reference (models/ddpm.py:314,327,337)."""
        if not self._on:
            return value
        v = value.detach().clone()
        if _staged(v, self.group):
            host = v.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            return host.to(v.device) / self.world
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        return v / self.world


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row range of rank `rank` when n_rows rows are split over `world` ranks
    (the sampler shards the initial noise this way; no communication is needed)."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def generate_sharded(model, initial_noise: torch.Tensor, diffusion_steps: int, group=None, gather=True):
    """Every rank denoises its own rows of `initial_noise`; with gather=True the rows are
    all-gathered at the end (the only collective, outside the sampling loop)."""
    rank, world = world_info(group)
    lo, hi = shard_rows(initial_noise.shape[0], rank, world)
    mine = model.generate(hi - lo, diffusion_steps, initial_noise=initial_noise[lo:hi]) if hi > lo else \
        initial_noise[:0]
    if world == 1 or not gather:
        return mine
    ranges = [shard_rows(initial_noise.shape[0], r, world) for r in range(world)]
    parts = [torch.empty((b - a,) + tuple(initial_noise.shape[1:]), dtype=mine.dtype, device=mine.device)
             for a, b in ranges]
    parts[rank] = mine.contiguous()
    for r in range(world):                      # rows per rank may differ by one: one broadcast each
        if parts[r].numel():
            src = dist.get_global_rank(group, r) if group is not None else r
            broadcast(parts[r], src, group)
    return torch.cat(parts, 0)
