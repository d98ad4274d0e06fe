#!/bin/bash
# Same-box A/B of two builds of the library on whole training steps: bash tests/ab_lib.sh "preset args" ...
# (base = vision_amd/lib/libvision_amd_base.so, built from an earlier commit by hand; new = the product library)
OUT=gpurun_out/ab_lib; mkdir -p $OUT
for spec in "$@"; do
    name=$(echo $spec | tr ' -' '__')
    for lib in base new base new; do
        L=$PWD/vision_amd/lib/libvision_amd.so; [ $lib = base ] && L=$PWD/vision_amd/lib/libvision_amd_base.so
        VAMD_LIB=$L timeout -k 10 300 python bench.py --preset $spec --no-cpu-baseline --no-sampling > $OUT/${name}_$lib.json 2> $OUT/${name}_$lib.err || { echo "FAILED $spec $lib"; tail -5 $OUT/${name}_$lib.err; exit 1; }
        python - <<PY
import json
d=json.load(open("$OUT/${name}_$lib.json"))
print("$lib $spec: %.3f ms/step %.1f img/s nt %.1f TF wgrad %.1f TF mfma_ms %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["wgrad_kernel"]["achieved_tflops"], d["mfma_conv_ms_per_step"]))
PY
    done
done
