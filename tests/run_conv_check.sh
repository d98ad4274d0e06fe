# kernel tests once per $TESTVARIANTS assignment (default: plain), then the conv micro-benchmark per $VARIANTS
# entry (an entry is one or more comma-separated NAME=value assignments)
for tv in ${TESTVARIANTS:-X=0}; do echo "== tests $tv"; env $tv timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -q --maxfail=300 --timeout 120 --timeout-method thread > gpurun_out/kernels.txt 2>&1; tail -3 gpurun_out/kernels.txt; grep FAILED gpurun_out/kernels.txt | head -20; done
CASES="${CASES:-64,224,224,64,64,3,1,0 64,112,112,64,64,3,1,1 64,224,224,64,64,3,2,0 64,112,112,128,128,3,1,0 64,56,56,256,256,3,1,0 64,28,28,512,512,3,1,0 64,14,14,512,512,3,1,0 64,14,14,512,512,1,1,0 64,112,112,64,128,1,1,0 64,56,56,128,128,3,1,1}"
for v in $VARIANTS; do echo "== $v"; env ${v//,/ } timeout -k 10 200 python -u tests/perf_conv_gpu.py $CASES 2>&1 | grep -v amdgpu.ids || exit 1; done
