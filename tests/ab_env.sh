#!/bin/bash
# Same-box comparison of values of one environment variable on whole training steps:
#   bash tests/ab_env.sh VAR "v1 v2 ..." "preset args" ...     (every value twice per preset, interleaved)
VAR=$1; VALS=$2; shift 2
OUT=gpurun_out/ab_env_$VAR; mkdir -p $OUT
for spec in "$@"; do
    name=$(echo $spec | tr ' -' '__')
    for rep in 1 2; do
    for v in $VALS; do
        env $VAR=$v timeout -k 10 300 python bench.py --preset $spec --no-cpu-baseline --no-sampling > $OUT/${name}_$v.json 2> $OUT/${name}_$v.err || { echo "FAILED $spec $v"; tail -5 $OUT/${name}_$v.err; exit 1; }
        python - <<PY
import json
d=json.load(open("$OUT/${name}_$v.json"))
print("$VAR=$v $spec: %.3f ms/step %.1f img/s nt %.1f TF wgrad %.1f TF mfma_ms %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["wgrad_kernel"]["achieved_tflops"], d["mfma_conv_ms_per_step"]))
PY
    done
    done
done
