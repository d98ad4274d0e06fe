#!/bin/bash
# Same-box A/B of two builds of the library on single convolution shapes: bash tests/ab_conv.sh "<perf_conv cases>" [PERF_EPI]
# (base = vision_amd/lib/libvision_amd_base.so, built from an earlier commit by hand; new = the product library)
CASES=$1; EPI=${2:-0}
for lib in base new; do
    L=$PWD/vision_amd/lib/libvision_amd.so; [ $lib = base ] && L=$PWD/vision_amd/lib/libvision_amd_base.so
    echo "== $lib (PERF_EPI=$EPI)"
    VAMD_LIB=$L PERF_EPI=$EPI timeout -k 10 200 python tests/perf_conv_gpu.py $CASES || exit 1
done
