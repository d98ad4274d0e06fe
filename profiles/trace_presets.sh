#!/bin/bash
# Runs on the GPU box (gpurun -- bash profiles/trace_presets.sh [preset ...]): rocprofv3 kernel trace + statistics of the
# training bench of each preset (default: butterfly flowers ffhq), into gpurun_out/trace_<preset>/.  With --one-stream: in the
# default mode backward's weight-gradient kernels run beside other kernels and a kernel's duration includes the sharing.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
PRESETS=${@:-butterfly flowers ffhq}
for p in $PRESETS; do
    OUT=$ROOT/gpurun_out/trace_$p
    rm -rf "$OUT"; mkdir -p "$OUT"
    STEPS="--steps 10 --warmup 3"
    [ "$p" = ffhq ] && STEPS="--steps 5 --warmup 2"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/bench.py --preset $p --no-cpu-baseline --no-sampling --one-stream $STEPS > $OUT/bench.log 2>&1 || { echo "trace of $p failed"; tail -5 $OUT/bench.log; exit 1; }
    find $OUT -name '*_kernel_trace.csv' -delete          # tens of MB; the statistics are what is kept
    echo "$p traced"
done
