#!/bin/bash
# Timing-only: what the epilogue's stores cost the persistent forward / input-gradient kernels.  Needs
# python vision_amd/build_ext.py --variant nostore -DVAMD_ABLATE=12 (results of that library are wrong by construction).
# CUS: compute units the kernels spread over (VAMD_NT_CUS), to tell a per-CU cost from a chip-wide one.
CASES="${CASES:-64,112,112,128,128,3,1,0 64,56,56,256,256,3,1,0 64,28,28,512,512,3,1,0 64,224,224,64,64,3,1,0}"
for cus in ${CUS:-256}; do
for epi in ${EPIS:-0 3}; do
  for lib in libvision_amd.so libvision_amd_nostore.so; do
    echo "== VAMD_NT_CUS=$cus PERF_EPI=$epi $lib"
    VAMD_NT_CUS=$cus PERF_EPI=$epi VAMD_LIB=$PWD/vision_amd/lib/$lib timeout -k 10 200 python -u tests/perf_conv_gpu.py $CASES 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
done
