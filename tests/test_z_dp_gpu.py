"""Two data-parallel ranks on ONE MI355X (both on cuda:0; gloo, device tensors staged through host memory): the
engine's backward issues the bucketed gradient all-reduce through GradReducer, FusedAdamW waits
for it, buffers are broadcast from rank 0.  Checked against a single-process run of the same two
per-rank batches: reduced gradients = mean of the per-rank gradients (fp32 compute, 1e-5), and
both ranks hold identical parameters after two optimizer steps.  RCCL itself is exercised by
`bench.py --gpus N` on a multi-GPU node; this test covers everything around the collective."""
import os
import socket
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

import torch.distributed as dist            # noqa: E402
import torch.multiprocessing as mp          # noqa: E402

from conftest import load_golden, sub       # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch(cfg, B, rank, step):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    P = cfg["patch_size"]
    return (torch.rand((B, 3, P, P), generator=g), torch.randn((B, 3, P, P), generator=g),
            torch.rand((B, 1, 1, 1), generator=g))


def _setup(kind):
    """(model on cuda:0 in fp32 compute with the fixture's initial state, config, batch, loss(model, batch tensors))."""
    import vision_amd
    dev = torch.device("cuda:0")
    if kind == "ddpm":
        t, cfg, B = load_golden("t1")
        state = sub(t, "state0")
        loss = lambda m, x, eps, tt: m.p_loss(x, noises=eps, diffusion_times=tt)
    else:                                   # the first stage: codebook gradient at arena offset 0, written mid-backward
        from test_vqunet_cpu import load_case
        cfg, t = load_case("v0")
        B = 4
        state = {k[len("state/"):]: v for k, v in t.items() if k.startswith("state/")}
        loss = lambda m, x, eps, tt: m._step_loss((x, None))
    model = vision_amd.get_model(kind, cfg, compute_dtype="fp32").to(dev).train()
    return model, state, cfg, B, loss


def _worker(rank, world, port, out_dir, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import vision_amd
    from vision_amd.parallel import DataParallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        model, state, cfg, B, loss_of = _setup(kind)
        if rank == 1:                      # rank 1 starts from different weights: the broadcast must fix that
            state = {k: (v + 0.25 if v.dtype.is_floating_point else v) for k, v in state.items()}
        model.load_state_dict(state)
        dp = DataParallel(model, bucket_bytes=64 << 10)          # several buckets even on the tiny model
        assert len(dp.reducer.buckets) > 2
        opt = model.configure_optimizers()[0][0]
        eng = model._ensure_engine()
        grads = []
        for step in range(2):
            x, eps, tt = (v.to(dev) for v in _batch(cfg, B, rank, step))
            dp.broadcast_buffers()
            loss = loss_of(model, x, eps, tt)
            loss.backward()
            dp.reducer.wait()
            grads.append(eng.flat_g.clone().cpu())
            opt.step()
        torch.cuda.synchronize()
        torch.save({"grads": grads, "params": eng.flat_p.cpu(), "bufs": eng.flat_b.cpu()},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["ddpm", "vqvaeunet"])
def test_two_ranks_match_single_process(tmp_path, kind):
    world = 2
    ctx = mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), kind), nprocs=world, join=False)
    deadline = time.time() + 90
    while not ctx.join(timeout=5):                     # never leave the suite hanging on a stuck rank
        if time.time() > deadline:
            for proc in ctx.processes:
                proc.kill()
            pytest.fail("data-parallel ranks did not finish within 90 s")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    # identical reduced gradients and parameters on both ranks
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)
    assert torch.equal(r0["params"], r1["params"])

    # single process: per-rank gradients of the same batches, averaged by hand
    dev = torch.device("cuda:0")
    model, state, cfg, B, loss_of = _setup(kind)
    model.load_state_dict(state)
    eng = model._ensure_engine()
    opt = model.configure_optimizers()[0][0]
    bufs0 = eng.flat_b.clone()
    for step in range(2):
        per_rank = []
        bufs_after_rank0 = None
        for rank in range(world):
            eng.flat_b.copy_(bufs0)                       # every rank starts the step from rank 0's buffers
            x, eps, tt = (v.to(dev) for v in _batch(cfg, B, rank, step))
            loss = loss_of(model, x, eps, tt)
            loss.backward()
            per_rank.append(eng.flat_g.clone())
            if rank == 0:
                bufs_after_rank0 = eng.flat_b.clone()
        mean = (per_rank[0] + per_rank[1]) / 2
        got = r0["grads"][step].to(dev)
        err = ((got - mean).norm() / mean.norm()).item()
        assert err < 1e-5, (step, err)
        eng.flat_g.copy_(mean)
        eng.flat_b.copy_(bufs_after_rank0)                # rank 0's statistics are what the next broadcast spreads
        bufs0 = bufs_after_rank0.clone()
        opt.step()
    # AdamW's first steps move every parameter by ~lr * sign(g): parameters whose exact gradient is
    # zero (their computed gradient is rounding noise that depends on summation order) may differ
    # by up to 2 * lr per step; everything else agrees to fp32 rounding
    diff = (r0["params"].to(dev) - eng.flat_p).abs()
    assert float(diff.max()) <= 2 * 2 * cfg["learning_rate"] * 1.01
    perr = (diff.norm() / eng.flat_p.norm()).item()
    assert perr < 2e-3, perr
    assert torch.allclose(r0["bufs"].to(dev), eng.flat_b, rtol=1e-5, atol=1e-6)


def test_rccl_path_with_one_rank_group():
    """bench.py under torch.distributed.run with ONE rank and VAMD_DP_FORCE=1: the parameter / buffer
    broadcasts and the bucketed asynchronous gradient all-reduce (ReduceOp.AVG) go through RCCL on
    this GPU, exactly as they do with N ranks; the step must still train (finite loss)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VAMD_DP_FORCE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-sampling", "--preset", "flowers",
           "--batch", "8"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0
    assert 0.0 < res["final_loss"] < 10.0
    ex = res["exchange"]          # the communicator counted its own ranks; the exchange ran and was timed on the device
    assert ex["backend"] == "nccl" and ex["rccl_ranks"] == 1 and ex["live"] and len(ex["buckets_bytes"]) >= 4
    assert ex["exposed_allreduce_ms_per_step"] is not None and 0.0 <= ex["exposed_allreduce_ms_per_step"] < 50.0


def test_run_py_with_two_ranks_fits_and_tests_on_rank0(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 run.py`: both ranks fit (sharded synthetic batches, gradient
    all-reduce, scalar all-reduce of the logged losses), then rank 0 ALONE runs the test pass with a trainer
    that takes no part in the process group while rank 1 waits in the final barrier (reference run.py:93-103).
    Two gloo ranks share this GPU; a rank-0-only collective would leave the job hanging until the timeout."""
    import json
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _, cfg, _ = load_golden("t1")
    config = {
        "model_config": cfg,
        "data_config": {"patch_size": cfg["patch_size"], "data_dir": "unused", "batch_size": 4, "length": 100},
        "trainer_config": {"accelerator": "gpu", "devices": 2, "max_epochs": 1, "precision": "32"},
        "log_config": {"name": "dp2", "save_dir": str(tmp_path / "logs"), "checkpoint": True, "model_summary": False},
    }
    path = tmp_path / "dp2.yaml"
    path.write_text(yaml.safe_dump(config, sort_keys=False))
    env = dict(os.environ, VAMD_DIST_BACKEND="gloo", VAMD_DIST_TIMEOUT_S="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "run.py"),
           "--config", str(path), "--synthetic"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "TeL" in out.stdout
    v0 = tmp_path / "logs" / "dp2" / "version_0"
    rows = [json.loads(l) for l in open(v0 / "metrics.jsonl")]
    assert len(rows) == 1 and rows[0]["step"] == 10         # 80 training images / 2 ranks / batch 4
    assert (v0 / "checkpoints" / "last.safetensors").exists()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself (as a child,
    before it touches the GPU) and relays rank 0's line, which must report the two ranks that joined, the buckets of the
    exchange and its exposed time per step.  Two gloo ranks share this one GPU (VAMD_DIST_BACKEND=gloo); with RCCL the
    same command needs two GPUs and must refuse to run here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
            "--no-sampling", "--preset", "flowers", "--batch", "4"]
    out = subprocess.run(args, env=dict(base, VAMD_DIST_BACKEND="gloo", VAMD_DIST_TIMEOUT_S="90"), cwd=root,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 8 and res["config"]["parallelism"] == "dp2"
    ex = res["exchange"]
    assert ex["ranks"] == 2 and ex["backend"] == "gloo" and ex["rccl_ranks"] is None and ex["live"]
    assert sum(ex["buckets_bytes"]) == ex["payload_bytes_per_step"] and 99e6 < ex["payload_bytes_per_step"] < 100e6
    assert ex["exposed_allreduce_ms_per_step"] is not None and ex["exposed_allreduce_ms_per_step"] >= 0.0
    assert 0.0 < res["final_loss"] < 10.0
    if torch.cuda.device_count() < 2:
        out = subprocess.run(args, env=base, cwd=root, capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and "{" not in out.stdout and "visible" in out.stderr
