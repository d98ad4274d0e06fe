#!/bin/bash
# A/B of one environment switch on the GPU box: bash tests/ab_bench.sh VAR "preset args" ...   (runs VAR=0 then VAR=1 per preset)
VAR=$1; shift
OUT=gpurun_out/ab_$VAR; mkdir -p $OUT
for spec in "$@"; do
    name=$(echo $spec | tr ' -' '__')
    for v in 0 1; do
        env $VAR=$v timeout -k 10 300 python bench.py --preset $spec --no-cpu-baseline --no-sampling > $OUT/${name}_$v.json 2> $OUT/${name}_$v.err || { echo "FAILED $spec $v"; tail -5 $OUT/${name}_$v.err; exit 1; }
        python - <<PY
import json
d=json.load(open("$OUT/${name}_$v.json"))
print("$VAR=$v $spec: %.3f ms/step %.1f img/s nt %.1f TF wgrad %.1f TF mfma_ms %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["wgrad_kernel"]["achieved_tflops"], d["mfma_conv_ms_per_step"]))
PY
    done
done
