"""Benchmark of the DDPM UNet training step (and sampler) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset butterfly|flowers|ffhq|vqvaeunet|ldm] [--batch B] [--graph]

`--graph`: the timed steps are replays of ONE captured training step (vision_amd.graphed.GraphedTrainStep: the host issues a
step in ~0.15 ms instead of 4-12 ms; same kernels, same order).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process, before it touches a GPU, starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child process
(one rank per GPU over RCCL, as the reference gets N ranks from trainer_config.devices, reference run.py:81-91),
relays rank 0's JSON line and exits with the child's status.  Fewer than N visible GPUs is an error, never a
one-GPU line.  Under an outer torch.distributed.run (WORLD_SIZE set) it is a rank of that job.

One "step" = one full training step of the hot path on one synthetic ImageFolder-shaped batch per
GPU: schedule + q_sample, UNet forward, L1 loss, hand-written backward, (bucketed RCCL gradient
all-reduce when N > 1), fused AdamW, weight re-pack.  Inputs are resident in HBM before the timed
region.  The workload at N = 1 is BASELINE.json configs[1]: configs/butterfly_ddpm.yaml,
batch 64 per GPU, 224x224, bf16 compute.  N > 1 is weak scaling (64 images per GPU).

Rank 0 prints ONE JSON line: metric value = training images/s over all GPUs, plus
  roofline     - the MFMA implicit-GEMM kernel (forward + input-gradient launches): algorithmic
                 FLOPs / device time, from HIP events the library records on the launch stream
                 right before and after every launch of that kernel, over up to 20 further steps
                 issued right after the timed ones (the ~220 event packets a step needs cost it
                 about 0.65 ms, so the K timed steps run without them; `ms_per_step_with_launch_events`
                 is the instrumented steps' time);
  cpu_baseline - the oracle's CPU restatement timed on a bounded sample of configs[0]
                 (flowers model, batch 8), N = 1 only.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.perf_counter()
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak of MI355X
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="butterfly")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the preset's data_config.batch_size)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true",
                    help="capture the training step into a HIP graph (vision_amd.graphed) and time its replays; the per-launch "
                         "roofline figures are then taken over 10 eager steps before the capture")
    ap.add_argument("--one-stream", action="store_true",
                    help="keep backward's weight-gradient kernels on the main stream in every step (kernel profiles in which "
                         "no two kernels overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampling", action="store_true")
    ap.add_argument("--sample-steps", type=int, default=None,
                    help="denoising steps of the timed sampling run (default 10; 1000 for --preset ffhq, as BASELINE names it)")
    return ap.parse_args()


def cpu_baseline(seconds_budget=90.0):
    """Oracle (plain-PyTorch CPU restatement of the reference) on BASELINE configs[0]: flowers model,
    batch 8, fp32: the 20 training steps the config names (the budget only guards against a host that is
    many times slower than the GPU box's 16 cores, where 20 steps take about 15 s), then its 10-step sampling
    of 4 images."""
    from oracle import restatement as R
    from vision_amd.presets import get_preset
    cfg = get_preset("flowers")["model_config"]
    sd = R.init_state_like_torch(cfg, cfg["manual_seed"])
    names = [k for k in sd if R.is_param(k)]
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    g = torch.Generator().manual_seed(0)
    B, P = 8, cfg["patch_size"]
    x = torch.rand((B, 3, P, P), generator=g)
    steps, t0 = 0, time.perf_counter()
    while True:
        eps = torch.randn((B, 3, P, P), generator=g)
        t = torch.rand((B, 1, 1, 1), generator=g)
        loss, grads = R.loss_and_grads(sd, cfg, x, eps, t)
        R.adamw_step({k: sd[k] for k in names}, grads, m, v, steps + 1, cfg["learning_rate"], cfg["weight_decay"])
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or steps >= 20:
            break
    train_ips = steps * B / el
    t1 = time.perf_counter()
    with torch.no_grad():
        R.reverse_diffusion(sd, cfg, torch.randn((4, 3, P, P), generator=g), 10)
    samp = time.perf_counter() - t1
    return {"value": round(train_ips, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "steps": steps, "batch": B, "sampling_images_per_s": round(4 / samp, 3), "sampling_image_steps_per_s": round(4 * 10 / samp, 2),
            "sample": f"flowers_ddpm model, batch 8, fp32, {steps} training steps ({el:.1f} s) on the host CPU; "
                      f"sampling 4 images x 10 steps took {samp:.1f} s"}


def cpu_baseline_latent(kind, seconds_budget=16.0):
    """Oracle on the host for the two latent-diffusion presets, fp32: the first stage's training step (224 x 224, batch 4,
    Adam with L2 decay), or the latent UNet's training step on the first stage's encoder output (8 images per step: encoder
    forward without gradient, then loss, gradients and AdamW of the latent UNet).  Steps until the budget is used."""
    from oracle import ldm as L
    from oracle import restatement as R
    from oracle import vqunet as V
    import vision_amd
    from vision_amd.presets import get_preset
    fs_cfg = get_preset("vqvaeunet")["model_config"]
    torch.manual_seed(fs_cfg["manual_seed"])
    fs = {k: v.clone() for k, v in vision_amd.get_model("vqvaeunet", fs_cfg).state_dict().items()}     # host tensors: no engine yet
    g = torch.Generator().manual_seed(0)
    P = fs_cfg["patch_size"]
    if kind == "vqvaeunet":
        B = 4
        x = torch.rand((B, 3, P, P), generator=g)
        params = {k: fs[k] for k in fs if V.is_param(k)}
        m = {k: torch.zeros_like(v) for k, v in params.items()}
        v2 = {k: torch.zeros_like(v) for k, v in params.items()}
        steps, t0 = 0, time.perf_counter()
        while True:
            grads = V.loss_and_grads(fs, fs_cfg, x)[-1]
            V.adam_l2_step(params, grads, m, v2, steps + 1, fs_cfg["learning_rate"], fs_cfg["weight_decay"])
            steps += 1
            el = time.perf_counter() - t0
            if el > seconds_budget or steps >= 20:
                break
        what = f"butterfly_vqvaeunet first stage, batch {B}, fp32, {steps} training steps ({el:.1f} s) on the host CPU"
    else:
        B = 8
        cfg = get_preset("ldm")["model_config"]
        torch.manual_seed(cfg["manual_seed"])
        sd = {k: v.clone() for k, v in vision_amd.get_model("ldm", cfg).state_dict().items()}
        names = [k for k in sd if L.is_param(k)]
        m = {k: torch.zeros_like(sd[k]) for k in names}
        v2 = {k: torch.zeros_like(sd[k]) for k in names}
        x = torch.rand((B, 3, P, P), generator=g)
        steps, t0 = 0, time.perf_counter()
        while True:
            with torch.no_grad():
                z, _ = V.encode(fs, fs_cfg, x, False)
            eps = torch.randn(z.shape, generator=g)
            t = torch.rand((B, 1, 1, 1), generator=g)
            loss, grads = L.loss_and_grads(sd, cfg, z, eps, t)
            L.p_loss(sd, cfg, z, eps, t, True)          # the statistics update of the step
            R.adamw_step({k: sd[k] for k in names}, grads, m, v2, steps + 1, cfg["learning_rate"], cfg["weight_decay"])
            steps += 1
            el = time.perf_counter() - t0
            if el > seconds_budget or steps >= 20:
                break
        what = (f"butterfly_ldm latent UNet on the butterfly_vqvaeunet encoder output, batch {B}, fp32, {steps} training steps "
                f"({el:.1f} s) on the host CPU")
    return {"value": round(steps * B / el, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "steps": steps, "batch": B, "sample": what}


TRAFFIC_CSV = "r3_pmc_tcc_traffic.csv"


def measured_traffic(*kernel_substrs):
    """HBM bytes per launch of the kernels whose name contains one of `kernel_substrs`, from the committed
    counter pass profiles/r3_pmc_tcc_traffic.csv ((2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, same
    workload).  The file's first line carries the hash of the kernel sources it was collected with: None when
    the file is missing or the kernels in the tree have changed since (counters cannot be collected inside
    a timed run, so a stale figure is dropped rather than reported)."""
    import csv
    from vision_amd._lib import source_hash
    path = os.path.join(ROOT, "profiles", TRAFFIC_CSV)
    if not os.path.exists(path):
        return None
    nbytes, launches = 0.0, 0
    with open(path) as f:
        stamp = f.readline().strip()
        if stamp != f"# kernel_source_sha256_16={source_hash()}":
            return None
        for row in csv.reader(f):
            if any(k in row[0] for k in kernel_substrs):
                nbytes += (2.0 * float(row[2]) + float(row[3])) * 1024.0
                launches += int(row[1])
    return round(nbytes / launches) if launches else None


def trace(msg):
    if os.environ.get("VAMD_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')} t={time.perf_counter() - T_START:7.2f}s] {msg}", file=sys.stderr, flush=True)


def launch_ranks(args):
    """`--gpus N` (N > 1) outside a torch.distributed.run job: start one as a CHILD process (this process has not touched
    a GPU and never replaces itself), pass rank 0's JSON line through and return the child's exit status.  The line
    must report the N ranks that were asked for; anything else is an error."""
    n = args.gpus
    backend = os.environ.get("VAMD_DIST_BACKEND", "nccl")
    visible = torch.cuda.device_count()               # counts devices without initialising one
    if visible < 1 or (backend == "nccl" and visible < n):
        print(f"bench.py: --gpus {n} asked for, {visible} GPU(s) visible: RCCL needs one GPU per rank "
              f"(VAMD_DIST_BACKEND=gloo lets several ranks share a GPU to rehearse the exchange path)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for out in child.stdout:
        if out.startswith("{"):
            line = out
        sys.stdout.write(out)
        sys.stdout.flush()
    rc = child.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank job ended with status {rc}", file=sys.stderr)
        return rc
    try:
        joined = json.loads(line)["n_gpus"] if line else None
    except (ValueError, KeyError):
        joined = None
    if joined != n:
        print(f"bench.py: asked for {n} ranks, the job reported {joined}", file=sys.stderr)
        return 3
    return 0


def main():
    args = parse()
    if args.sample_steps is None:
        args.sample_steps = 1000 if args.preset == "ffhq" else 10
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if os.environ.get("VAMD_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(45, exit=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a line that says n_gpus = WORLD_SIZE for --gpus N would be read as the N-GPU result
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU; VAMD_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path
    backend = os.environ.get("VAMD_DIST_BACKEND", "nccl")
    local_rank = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # VAMD_DP_FORCE=1 with WORLD_SIZE=1 (under torch.distributed.run): a one-rank process group, so
    # that the exchange path runs through RCCL on a one-GPU machine
    grouped = world > 1 or (os.environ.get("VAMD_DP_FORCE") == "1" and "MASTER_ADDR" in os.environ)
    if grouped:
        # a collective that some rank never joins ends the run with an error instead of hanging it
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("VAMD_DIST_TIMEOUT_S", "300"))))
    rank = dist.get_rank() if grouped else 0
    joined = dist.get_world_size() if grouped else 1
    # ranks that take part in the communicator, counted by the communicator itself: a sum of ones over RCCL
    rccl_ranks = None
    if grouped and backend == "nccl":
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: WORLD_SIZE={world} but {rccl_ranks} ranks answered the RCCL all-reduce")

    trace("process group up")
    import vision_amd
    from vision_amd.parallel import DataParallel
    from vision_amd._lib import lib
    from vision_amd.presets import get_preset

    preset = get_preset(args.preset)
    cfg = preset["model_config"]
    kind = cfg["name"]                   # 'ddpm' | 'vqvaeunet' (first stage) | 'ldm' (latent diffusion over the first stage)
    B = args.batch or preset["data_config"]["batch_size"]
    P = preset["data_config"]["patch_size"]
    torch.manual_seed(cfg["manual_seed"])
    model = vision_amd.get_model(kind, cfg, compute_dtype=args.dtype).to(dev).train()
    if kind == "ldm":
        # image batches in, the first stage's encoder inside the timed step (frozen, random-init like everything here)
        fs_cfg = get_preset("vqvaeunet")["model_config"]
        model.set_first_stage(vision_amd.get_model("vqvaeunet", fs_cfg, compute_dtype=args.dtype).to(dev).eval())
    trace("model on device")
    dp = DataParallel(model)
    trace("parameters and buffers broadcast")
    eng = model._ensure_engine()
    if args.one_stream:
        eng.side_wgrad = False
    opt = model.configure_optimizers()[0][0]

    gen = torch.Generator(device="cpu").manual_seed(rank)
    nbatch = 2
    batches = [(torch.rand((B, 3, P, P), generator=gen).to(dev), torch.zeros(B, dtype=torch.int64, device=dev))
               for _ in range(nbatch)]

    stepper = None

    ahead = getattr(model, "encode_ahead", None)      # latent diffusion: the next batch's first-stage encoding starts now,
                                                      # as vision_amd.trainer's loop does (it runs beside this step)
    def step(i):
        dp.broadcast_buffers()
        if stepper is not None:
            return stepper(batches[i % nbatch], i)
        if ahead is not None:
            ahead(batches[(i + 1) % nbatch])
        loss = model.training_step(batches[i % nbatch], i)
        loss.backward()
        opt.step()
        return loss

    trace("model built")
    for i in range(args.warmup):
        step(i)
        trace(f"warmup step {i} issued")
    if hasattr(model, "train_losses"):
        model.train_losses.clear()

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    trace("warmup done")
    import ctypes
    if args.graph:
        from vision_amd.graphed import GraphedTrainStep
        stepper = GraphedTrainStep(model, opt)
        for i in range(3):                      # two eager steps, then the capture and its first replay
            step(args.warmup + i)
        fence()
        trace("graph captured")
    dp.reducer.timing = []        # (event after the last backward kernel, event after the last bucket has landed) per step
    # ---- the timed steps.  The library's per-launch events are OFF here: two event packets around each of a step's ~110
    # MFMA launches cost the step about 0.65 ms (3.5 % of the butterfly step, 9 % of the flowers step, measured against
    # the same steps without them), so the roofline figures are taken over the steps that follow, not over these.
    # Host time to issue a step: taken over the first steps after the fence, while the device queue still has room.
    # (The host may run ahead of the device by a bounded number of launches, about four steps' worth here; past that
    # every launch call blocks until the device has retired one, and the host's pace is the device's.)
    lib.vamd_launch_timing(0)
    n_free = min(4, args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        if i + 1 == n_free:
            host_free = (time.perf_counter() - t0) / n_free
        trace(f"timed step {i} issued")
    fence()
    trace("timed steps done")
    elapsed = time.perf_counter() - t0
    if grouped:
        tmax = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    exposed = dp.reducer.exposed_ms()
    dp.reducer.timing = None
    ips = world * B * args.steps / elapsed
    final_loss = float(loss.detach())

    # ---- the same steps again (launch by launch, also after --graph) with HIP events recorded inside the library on the
    # launch stream, directly around every MFMA convolution launch: the roofline of the dominant kernel.
    # VAMD_BENCH_LAYERS=1 adds per-call events in the engine for a per-layer table.
    # In the timed steps backward's weight-gradient kernels run on a second stream, beside the input-gradient chain and
    # its normalisation passes; a kernel timed while another one shares the chip says little about the kernel.  So the
    # instrumented steps keep everything on one stream (kernels one at a time), and a few steps of that kind WITHOUT the
    # events are timed first: `ms_per_step_one_stream` beside `ms_per_step` is what the second stream buys.
    layers = bool(os.environ.get("VAMD_BENCH_LAYERS"))
    prof = []                 # (kind, flops, start_event, end_event, shape) per MFMA conv call
    n_inst = min(args.steps, 20)
    stepper = None
    two_streams = bool(getattr(eng, "side_wgrad", False))
    one_stream_ms = None
    if two_streams:
        eng.side_wgrad = False
        step(args.warmup + args.steps)
        fence()
        t1 = time.perf_counter()
        for i in range(n_inst):
            step(args.warmup + args.steps + i)
        fence()
        one_stream_ms = 1e3 * (time.perf_counter() - t1) / n_inst
    lib.vamd_launch_timing(1)
    t1 = time.perf_counter()
    for i in range(n_inst):
        eng.profile = prof if layers else None
        step(args.warmup + args.steps + i)
    fence()
    inst_ms = 1e3 * (time.perf_counter() - t1) / n_inst
    eng.profile = None
    lib.vamd_launch_timing(0)
    if two_streams:
        eng.side_wgrad = True

    def timed(kind):
        ms, fl, ex = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = lib.vamd_launch_timing_read(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ex))
        if n < 0:
            raise RuntimeError(lib.vamd_last_error().decode())
        return fl.value, ms.value, n, ex.value

    roof = None
    extra = {}
    conv_ms = 0.0
    f, t, n, fx = timed(0)
    if n:
        conv_ms += t
        ach = f / (t * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_ntw_mfma_kernel / conv_ntp_mfma_kernel / conv_c64_halo_kernel <bf16> (persistent implicit-GEMM, forward + input gradient)",
                "achieved": round(ach, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
                "traffic": measured_traffic("conv_ntp_mfma_kernel", "conv_ntw_mfma_kernel", "conv_c64_halo_kernel") if args.preset == "butterfly" and B == 64 else None,
                "traffic_unit": f"HBM bytes per launch (PMC pass of the same command, profiles/{TRAFFIC_CSV})",
                "launches": n, "avg_launch_ms": round(t / n, 5), "flops_per_launch": round(f / n),
                # operator FLOPs above; what the launches really multiplied (the combined-filter forms of the
                # upsampling convolution execute 16/36 of the operator's count) beside it
                "executed_tflops": round(fx / (t * 1e-3) / 1e12, 1),
                "executed_frac": round(fx / (t * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}
    if roof is not None:
        roof["measured_over"] = (f"{n_inst} steps issued right after the timed ones with the library's per-launch HIP events on "
                                 f"({inst_ms:.3f} ms per step with the events)"
                                 + (" and every kernel on one stream (in the timed steps the weight-gradient kernels run on a "
                                    "second stream beside these launches)" if two_streams else ""))
    f, t, n, fx = timed(1)
    if n:
        conv_ms += t
        extra["wgrad_kernel"] = {"kernel": "conv_tn_mfma_kernel<bf16> (weight gradient)", "achieved_tflops": round(f / (t * 1e-3) / 1e12, 1),
                                 "launches": n, "avg_launch_ms": round(t / n, 5)}
    if layers and rank == 0:
        per_shape = {}
        for _launch_kind, flops, e0, e1, shape in prof:
            f0, t0_, n0 = per_shape.get(shape, (0.0, 0.0, 0))
            per_shape[shape] = (f0 + flops, t0_ + e0.elapsed_time(e1), n0 + 1)
        for shape, (f, t, n) in sorted(per_shape.items(), key=lambda kv: -kv[1][1]):
            print("layer %-52s calls %3d  %7.3f ms/call  %7.1f TFLOP/s  %5.1f%% of MFMA conv time" % (
                str(shape), n, t / n, f / (t * 1e-3) / 1e12, 100 * t / sum(v[1] for v in per_shape.values())), file=sys.stderr)
    extra["mfma_conv_ms_per_step"] = round(conv_ms / n_inst, 3)
    extra["ms_per_step_with_launch_events"] = round(inst_ms, 3)
    extra["peak_hbm_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)      # torch's allocator, training steps only so far
    extra["weight_gradient_stream"] = two_streams
    if one_stream_ms is not None:
        extra["ms_per_step_one_stream"] = round(one_stream_ms, 3)
    # host time to issue a step's launches vs the step's wall time: the step is device-bound while the first
    # stays below the second
    extra["host_issue_ms_per_step"] = round(1e3 * host_free, 3)
    # the exchange step: what the gradient all-reduce costs the compute stream per step AFTER backward's last kernel
    # (everything before that overlaps with backward), the buckets it moves, and who took part
    red = dp.reducer
    extra["exchange"] = {
        "backend": (backend if grouped else None), "ranks": joined, "rccl_ranks": rccl_ranks,
        "live": bool(red._on), "buckets_bytes": [4 * (hi - lo) for lo, hi in red.buckets] if red._on else [],
        "payload_bytes_per_step": 4 * red.total if red._on else 0,
        "exposed_allreduce_ms_per_step": (round(sum(exposed) / len(exposed), 4) if exposed else None),
        "exposed_allreduce_ms_max": (round(max(exposed), 4) if exposed else None),
    }

    sampling = None
    if not args.no_sampling and kind != "vqvaeunet":
        model.eval()
        nimg = min(B, 16)
        shape = (nimg, 3, P, P) if kind == "ddpm" else (nimg, cfg["h_dims"][0], cfg["in_res"], cfg["in_res"])
        noise = torch.randn(shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1 + rank))
        model.generate(nimg, 2, initial_noise=noise)
        fence()
        t1 = time.perf_counter()
        model.generate(nimg, args.sample_steps, initial_noise=noise)
        fence()
        ts = time.perf_counter() - t1
        sampling = {"images_per_s": round(world * nimg / ts, 3), "diffusion_steps": args.sample_steps,
                    "images_per_gpu": nimg, "image_steps_per_s": round(world * nimg * args.sample_steps / ts, 2)}

    if rank == 0:
        from vision_amd import flops
        if kind == "ddpm":
            tf = flops.train_flops_per_image(cfg)
            metric = "DDPM UNet train images/s (butterfly_ddpm UNet, 224x224, bf16)" if args.preset == "butterfly" \
                else f"DDPM UNet train images/s ({args.preset}_ddpm)"
            what = f"configs/{args.preset}_ddpm.yaml UNet training step"
        elif kind == "vqvaeunet":
            tf = flops.first_stage_train_flops_per_image(cfg)
            metric = "VQ-VAE-UNet first stage train images/s (butterfly_vqvaeunet)"
            what = "configs/butterfly_vqvaeunet.yaml first-stage training step"
        else:
            # the latent UNet's step plus the first stage's encoder forward (a third of its forward + backward, halved again:
            # the decoder half does not run) -- counted as the latent UNet alone, the encoder's work is not credited
            tf = flops.latent_train_flops_per_image(cfg)
            metric = "Latent diffusion train images/s (butterfly_ldm over the butterfly_vqvaeunet first stage)"
            what = "configs/butterfly_ldm.yaml latent UNet training step on the encoder output of configs/butterfly_vqvaeunet.yaml"
        out = {
            "metric": metric,
            "value": round(ips, 2), "unit": "images/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{what}, per-GPU batch {B}, "
                                   f"{P}x{P}, synthetic ImageFolder-shaped images, random-init weights (seed 42)",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "train_tflops_per_s": round(ips * tf / 1e12, 1),
            "train_frac_of_bf16_mfma_peak": round(ips * tf / 1e12 / (BF16_MFMA_PEAK_TFLOPS * world), 4),
            "final_loss": round(final_loss, 5),
            "graph_replay": bool(args.graph),
            "roofline": roof, "sampling": sampling, **extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline() if kind == "ddpm" else cpu_baseline_latent(kind)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
