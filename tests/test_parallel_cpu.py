"""World-size-2 checks of the data-parallel layer on CPU tensors over gloo: the bucketed gradient
mean all-reduce issued while "backward" is still walking the arena, buffer / parameter
broadcast semantics, scalar logging reduction, and row sharding of the sampler.  The same code
runs over nccl (RCCL) on GPUs; only the tensors' device differs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision_amd.parallel import GradReducer, generate_sharded, shard_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(fn, world=2, *args):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
def _reducer_worker(rank, world, total, boundaries, bucket_bytes):
    g = torch.Generator().manual_seed(1234)
    per_rank = [torch.randn(total, generator=g) for _ in range(world)]
    want = torch.stack(per_rank).mean(0)
    flat = torch.zeros(total)
    red = GradReducer(flat, boundaries, bucket_bytes=bucket_bytes)
    assert red.buckets[-1][0] == 0 and red.buckets[0][1] == total
    assert all(a[0] == b[1] for a, b in zip(red.buckets, red.buckets[1:]))      # contiguous, tail first
    cuts = sorted(set(boundaries + [0, total]))
    for step in range(2):                                                        # reused across steps
        flat.zero_()
        red.begin()
        # "backward": layers finish from the arena's tail to its head
        for lo, hi in reversed(list(zip(cuts, cuts[1:]))):
            flat[lo:hi] = per_rank[rank][lo:hi] * (step + 1)
            before = red._next
            red.layer_done(lo)
            # a bucket is only ever issued once all of its slice is final
            assert all(b_lo >= lo for b_lo, _ in red.buckets[before:red._next])
        red.wait()
        assert red._next == len(red.buckets) and not red._pending
        torch.testing.assert_close(flat, want * (step + 1), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("bucket_bytes", [64, 4096, 1 << 30])
def test_grad_reducer_mean_allreduce_world2(bucket_bytes):
    boundaries = [0, 7, 40, 41, 300, 777, 1000]
    _spawn(_reducer_worker, 2, 1111, boundaries, bucket_bytes)


def _early_issue_worker(rank, world):
    """Buckets whose slice is complete are in flight before backward has finished."""
    total, boundaries = 400, [0, 100, 200, 300]
    flat = torch.full((total,), float(rank + 1))
    red = GradReducer(flat, boundaries, bucket_bytes=4 * 100)
    assert red.buckets == [(300, 400), (200, 300), (100, 200), (0, 100)]
    red.begin()
    red.layer_done(350)                 # inside the last layer: nothing complete yet
    assert red._next == 0
    red.layer_done(300)
    assert red._next == 1 and len(red._pending) == 1
    red.layer_done(100)
    assert red._next == 3
    red.wait()                          # issues the remaining bucket and drains
    assert torch.allclose(flat, torch.full((total,), 1.5))


def test_grad_reducer_issues_buckets_during_backward():
    _spawn(_early_issue_worker, 2)


class _FakeSampler:
    """Stands in for DDPMModel.generate: deterministic function of the rows it is given."""

    def generate(self, n, steps, initial_noise=None):
        assert initial_noise.shape[0] == n
        return initial_noise * steps + 1.0


def _sharded_worker(rank, world, rows):
    noise = torch.arange(rows * 6, dtype=torch.float32).reshape(rows, 1, 2, 3)
    out = generate_sharded(_FakeSampler(), noise, 3, gather=True)
    assert torch.equal(out, noise * 3 + 1.0)
    lo, hi = shard_rows(rows, rank, world)
    mine = generate_sharded(_FakeSampler(), noise, 3, gather=False)
    assert torch.equal(mine, noise[lo:hi] * 3 + 1.0)


@pytest.mark.parametrize("rows", [1, 2, 5, 8])
def test_generate_sharded_world2(rows):
    _spawn(_sharded_worker, 2, rows)


def _broadcast_worker(rank, world):
    """What DataParallel does with the engine's arenas: parameters and BatchNorm buffers of
    rank 0 replace everybody's; per-rank scalars are averaged for logging."""
    from vision_amd.parallel import DataParallel

    class _Eng:
        def __init__(self):
            self.flat_p = torch.full((10,), float(rank))
            self.flat_g = torch.zeros(10)
            self.flat_b = torch.full((4,), float(10 + rank))
            self.flat_nbt = torch.full((2,), 5 + rank, dtype=torch.int64)
            self.dirty = False
            self.reducer = None

        def layer_offsets(self):
            return [0, 4, 8]

    class _Model:
        def __init__(self):
            self.eng = _Eng()

        def _ensure_engine(self):
            return self.eng

    m = _Model()
    dp = DataParallel(m, bucket_bytes=16)
    assert torch.equal(m.eng.flat_p, torch.zeros(10)) and m.eng.dirty
    assert torch.equal(m.eng.flat_b, torch.full((4,), 10.0))
    assert torch.equal(m.eng.flat_nbt, torch.full((2,), 5, dtype=torch.int64))
    assert m.grad_reducer is dp.reducer and m.eng.reducer is dp.reducer
    v = dp.all_reduce_scalar_mean(torch.tensor(float(rank)))
    assert abs(float(v) - 0.5) < 1e-7


def test_data_parallel_broadcasts_world2():
    _spawn(_broadcast_worker, 2)


def test_shard_rows_partition():
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_grad_reducer_bucket_layout():
    """Buckets tile the arena from the tail, and the one that is complete only when backward ends
    (it starts at offset 0) is cut down to the first layers within head_bytes."""
    flat = torch.zeros(1000)
    bounds = [0, 10, 50, 100, 400, 700]
    for bucket_bytes, head_bytes in [(1200, 250), (1200, 20), (4000, 250), (4, 4), (1 << 20, 1 << 20)]:
        r = GradReducer(flat, bounds, bucket_bytes=bucket_bytes, head_bytes=head_bytes)
        assert r.buckets[0][1] == 1000 and r.buckets[-1][0] == 0
        for (lo, hi), (lo2, hi2) in zip(r.buckets, r.buckets[1:]):
            assert lo < hi and hi2 == lo                     # contiguous, tail first
        assert all(lo in bounds for lo, _ in r.buckets)      # cut at layer boundaries only
    r = GradReducer(flat, bounds, bucket_bytes=1200, head_bytes=250)
    assert r.buckets[-1] == (0, 50) and r.buckets[-2] == (50, 100)
    r = GradReducer(flat, bounds, bucket_bytes=1200, head_bytes=20)       # no boundary within the head: unchanged
    assert r.buckets[-1] == (0, 100)


def _timing_worker(rank, world):
    flat = torch.full((1000,), float(rank + 1))
    red = GradReducer(flat, [0, 250, 500, 750], bucket_bytes=1000)
    red.timing = []
    for _ in range(3):
        flat.fill_(float(rank + 1))
        red.begin()
        red.finish()
        red.wait()
        assert torch.allclose(flat, torch.full((1000,), 1.5))
    ms = red.exposed_ms()
    assert len(ms) == 3 and all(0.0 <= v < 10_000.0 for v in ms)          # one exposed-wait figure per step
    red.timing = None
    red.begin(); red.finish(); red.wait()
    assert red.exposed_ms() == []


def test_grad_reducer_reports_exposed_wait_per_step():
    _spawn(_timing_worker, 2)


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus 8` where fewer GPUs are visible ends with an error and prints no result line: a one-GPU line under
    that command would be read as the 8-GPU result."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VAMD_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "{" not in r.stdout and "8" in r.stderr and "visible" in r.stderr
    # an outer launcher whose world size disagrees with --gpus is refused as well
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "{" not in r.stdout and "WORLD_SIZE=2" in r.stderr
