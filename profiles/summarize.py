"""Turns gpurun_out/collect (written by profiles/collect.sh) into the committed summaries:
r2_train_step_kernel_stats.csv, r2_train_step_bench_line.json, r2_bench_lines.json (the three presets without a
profiler attached), r2_pmc_tcc_traffic.csv (first line: hash of the kernel sources it was taken with) and
r2_pmc_sq_summary.csv."""
import collections
import csv
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROUND = "r2"
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
    shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(HERE, ROUND + "_train_step_kernel_stats.csv"))
    line = [ln for ln in open(os.path.join(SRC, "trace.log")) if ln.startswith("{")][-1]
    open(os.path.join(HERE, ROUND + "_train_step_bench_line.json"), "w").write(line)

    sq, ns = counters("sq")
    names = sorted({c for v in sq.values() for c in v})
    with open(os.path.join(HERE, ROUND + "_pmc_sq_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches"] + names)
        for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            w.writerow([k, ns[k]] + [round(v.get(c, 0.0)) for c in names])
    with open(os.path.join(HERE, ROUND + "_bench_lines.json"), "w") as f:
        for preset in ("butterfly", "flowers", "ffhq"):
            f.write([ln for ln in open(os.path.join(SRC, f"bench_{preset}.json")) if ln.startswith("{")][-1])
    print("wrote summaries from", SRC)


if __name__ == "__main__":
    # --traffic-only: run on the GPU box right after the counter passes, so that the bench lines taken next in the same
    # collection find a traffic file stamped with the kernels they run
    if "--traffic-only" in sys.argv:
        write_traffic()
    else:
        write_traffic()
        write_rest()
