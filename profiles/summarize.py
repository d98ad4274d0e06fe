"""Turns gpurun_out/collect and gpurun_out/trace_<preset> (written by profiles/collect.sh) into the committed summaries:
r3_bench_lines.json (the five presets and the graph-replayed butterfly step, no profiler attached),
r3_pmc_tcc_traffic.csv (first line: hash of the kernel sources it was taken with), r3_pmc_sq_summary.csv, and per preset
r3_<preset>_kernel_stats.csv + r3_<preset>_bench_line_profiled.json with the per-family tables of r3_kernel_families.md
(--families writes those alone, from a run of profiles/trace_presets.sh)."""
import collections
import csv
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROUND = "r3"
SRC = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(HERE, "..", "gpurun_out", "collect")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    assert hits, pattern
    return hits[-1]                      # newest: results of earlier collections may still lie beside it


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(one(f"{sub}/*/*_counter_collection.csv"))):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


def write_traffic():
    fetch, nf = counters("fetch")
    write, _ = counters("write")
    stamp = open(os.path.join(SRC, "source_hash.txt")).read().strip()
    with open(os.path.join(HERE, ROUND + "_pmc_tcc_traffic.csv"), "w", newline="") as f:
        f.write(f"# kernel_source_sha256_16={stamp}\n")
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_sum", "WRITE_SIZE_KB_sum", "hbm_bytes_per_launch(2*FETCH+WRITE)*1024"])
        rows = []
        for k in fetch:
            fs, ws, n = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", 0.0), nf[k]
            rows.append((k, n, round(fs), round(ws), round((2 * fs + ws) * 1024 / n)))
        for row in sorted(rows, key=lambda r: -r[1] * r[4]):
            w.writerow(row)



def write_rest():
    sq, ns = counters("sq")
    names = sorted({c for v in sq.values() for c in v})
    with open(os.path.join(HERE, ROUND + "_pmc_sq_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches"] + names)
        for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            w.writerow([k, ns[k]] + [round(v.get(c, 0.0)) for c in names])
    with open(os.path.join(HERE, ROUND + "_bench_lines.json"), "w") as f:
        for preset in ("butterfly", "flowers", "ffhq", "vqvaeunet", "ldm", "butterfly_graph", "butterfly_one_stream"):
            f.write([ln for ln in open(os.path.join(SRC, f"bench_{preset}.json")) if ln.startswith("{")][-1])
    write_family_tables(os.path.join(SRC, ".."))
    print("wrote summaries from", SRC)


FAMILIES = [   # (family, kernel-name substrings, bound)
    ("forward / input-gradient convolutions (MFMA)", ("conv_ntw", "conv_ntp", "conv_nt8", "conv_c64_halo", "conv_nt_mfma", "splitk_reduce"), "mfma"),
    ("weight gradient (MFMA) + its reduce kernels", ("conv_tn", "wgrad_", "slab_sum", "column_sums", "unpack_wgrad"), "mfma"),
    ("normalise backward: apply", ("norm_bwd_apply",), "hbm"),
    ("normalise backward: statistics", ("chan_stats_kernelIDF16bLi8ELi1", "chan_stats_kernelIDF16bLi8ELi2", ", 1>", ", 2>"), "hbm"),
    ("normalise forward: statistics not fused into a convolution", ("chan_stats",), "hbm"),
    ("normalise forward: scale, shift, LeakyReLU", ("scale_shift_act",), "hbm"),
    ("normalisation coefficients (finalize kernels)", ("finalize", "fold_partials", "gn_bwd_coef", "sum_over_g"), "latency"),
    ("thin-channel convolutions (3 <-> 64 channels)", ("thin_", "tiny33"), "hbm"),
    ("attention core (products, softmax, transposes)", ("bmm", "softmax", "transpose", "attention"), "mfma"),
    ("time embedding", ("linear_", "timeembed"), "latency"),
    ("AdamW", ("adamw", "adam_l2"), "hbm"),
    ("weight re-pack", ("repack",), "hbm"),
    ("objective and layout (q_sample, L1, converters)", ("qsample", "l1_", "nchw", "nhwc", "schedule", "denoise"), "hbm"),
    ("torch (allocator fills, copies, random numbers)", ("at::native", "rocclr", "distribution"), "latency"),
]


def family_table(preset, stats_csv, bench_line, steps_traced):
    """Per kernel family of one training step: launches, time, share of the step's kernel time, and the achieved rate
    against the family's bound where the algorithmic work is defined: MFMA families from the FLOPs bench.py counted
    (operator FLOPs), HBM families from the bytes the pass must move (profiles/README.md says which)."""
    import json
    sys.path.insert(0, os.path.join(HERE, ".."))
    from vision_amd import flops
    from vision_amd.presets import get_preset
    line = json.loads(bench_line)
    cfg = get_preset(preset)["model_config"]
    B = line["config"]["global_batch"]
    bn, gn = flops.norm_elems_per_image(cfg)
    params = None
    rows = list(csv.DictReader(open(stats_csv)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = collections.OrderedDict((f[0], [0, 0.0]) for f in FAMILIES)
    fam["other"] = [0, 0.0]
    for r in rows:
        name = r["Name"]
        for label, keys, _ in FAMILIES:
            if any(k in name for k in keys):
                break
        else:
            label = "other"
        fam[label][0] += int(r["Calls"])
        fam[label][1] += float(r["TotalDurationNs"])
    roof, wg = line["roofline"], line["wgrad_kernel"]
    work = {   # family -> (algorithmic work per step, unit)
        FAMILIES[0][0]: (roof["flops_per_launch"] * roof["launches"] / line["steps"], "flop"),
        FAMILIES[1][0]: (wg["achieved_tflops"] * 1e12 * wg["avg_launch_ms"] * 1e-3 * wg["launches"] / line["steps"], "flop"),
        FAMILIES[2][0]: (3 * 2.0 * (bn + gn) * B, "byte"),     # dy and x read, dx written (bf16); merged addends not counted
        FAMILIES[3][0]: (2 * 2.0 * (bn + gn) * B, "byte"),     # dy and x read
        FAMILIES[5][0]: (2 * 2.0 * (bn + gn) * B, "byte"),     # x read, activation written
    }
    out = [f"### {preset}: kernel families of one training step (rocprofv3 kernel statistics over {steps_traced} steps; "
           f"bench line of the same run: {line['ms_per_step']} ms per step under the profiler)", "",
           "| family | launches / step | us / step | share | bound | achieved | of peak |", "|---|---|---|---|---|---|---|"]
    for label, (calls, ns) in fam.items():
        if calls == 0:
            continue
        us = ns / steps_traced / 1e3
        bound = next((b for l, _, b in FAMILIES if l == label), "-")
        ach = peak = ""
        if label in work and ns > 0:
            w, unit = work[label]
            if unit == "flop":
                rate = w / (us * 1e-6) / 1e12
                ach, peak = f"{rate:.0f} TFLOP/s", f"{rate / 2500.0:.2f}"
            else:
                rate = w / (us * 1e-6) / 1e12
                ach, peak = f"{rate:.2f} TB/s", f"{rate / 8.0:.2f}"
        out.append(f"| {label} | {calls / steps_traced:.1f} | {us:.0f} | {100 * ns / total:.1f} % | {bound} | {ach} | {peak} |")
    out.append(f"| all kernels | {sum(c for c, _ in fam.values()) / steps_traced:.1f} | {total / steps_traced / 1e3:.0f} | 100 % | | | |")
    return "\n".join(out) + "\n"


def write_family_tables(root):
    """root/trace_<preset>/ (profiles/trace_presets.sh) -> profiles/r3_<preset>_kernel_stats.csv and one table file."""
    tables = []
    for preset in ("butterfly", "flowers", "ffhq"):
        d = os.path.join(root, "trace_" + preset)
        hits = sorted(glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if not hits:
            continue
        dst = os.path.join(HERE, f"{ROUND}_{preset}_kernel_stats.csv")
        shutil.copy(hits[-1], dst)
        line = [ln for ln in open(os.path.join(d, "bench.log")) if ln.startswith("{")][-1]
        with open(os.path.join(HERE, f"{ROUND}_{preset}_bench_line_profiled.json"), "w") as f:
            f.write(line)
        import json
        d = json.loads(line)
        # the profiled process runs the warm-up steps, the timed steps and the instrumented steps that follow them
        steps = d["warmup"] + d["steps"] + min(d["steps"], 20) + (3 if d.get("graph_replay") else 0)
        if d.get("weight_gradient_stream"):
            steps += 1 + min(d["steps"], 20)            # the one-stream steps timed before the instrumented ones
        tables.append(family_table(preset, dst, line, steps))
    with open(os.path.join(HERE, f"{ROUND}_kernel_families.md"), "w") as f:
        f.write("\n".join(tables))
    print("".join(tables))


if __name__ == "__main__":
    if "--families" in sys.argv:
        write_family_tables(os.path.join(HERE, "..", "gpurun_out"))
        sys.exit(0)
    # --traffic-only: run on the GPU box right after the counter passes, so that the bench lines taken next in the same
    # collection find a traffic file stamped with the kernels they run
    if "--traffic-only" in sys.argv:
        write_traffic()
    else:
        write_traffic()
        write_rest()
