# correctness of the ping-pong kernel (whole tiles and split tiles, forced onto every shape it accepts), then the
# conv micro-benchmark without it, with whole tiles, and with split tiles
CASES="${CASES:-64,112,112,128,128,3,1,0 64,56,56,256,256,3,1,0 64,28,28,512,512,3,1,0 64,56,56,128,256,3,1,0 64,28,28,256,512,3,1,0 64,28,28,512,256,3,1,0 64,56,56,256,128,3,1,0 64,112,112,128,128,3,2,0 64,14,14,512,512,3,1,0 64,28,28,256,256,3,1,0 64,14,14,512,512,3,1,1 16,112,112,128,128,3,1,0 16,56,56,256,256,3,1,0}"
VAMD_NT8_MIN=1 VAMD_NT8_TERMS=1 VAMD_NT8_WIDE_K=1000000 timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/nt8_kernels.log 2>&1; rc=$?; tail -4 gpurun_out/nt8_kernels.log; [ $rc -eq 0 ] || exit $rc
VAMD_NT8_STREAMK=1 VAMD_NT8_TERMS=1 VAMD_NT8_WIDE_K=1000000 timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread -k wide_tiles > gpurun_out/nt8_kernels_split0.log 2>&1 && VAMD_NT8_STREAMK=1 VAMD_TEST_SCRATCH=1 VAMD_NT8_MIN=1 VAMD_NT8_MIN_SPLIT=1 VAMD_NT8_TERMS=1 VAMD_NT8_WIDE_K=1000000 timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/nt8_kernels_split.log 2>&1; rc=$?; tail -4 gpurun_out/nt8_kernels_split.log; [ $rc -eq 0 ] || exit $rc
[ -n "$SKIP_PERF" ] && exit 0
echo "== VAMD_NT8=0"; VAMD_NT8=0 timeout -k 10 200 python -u tests/perf_conv_gpu.py $CASES 2>&1 | grep -v amdgpu.ids || exit 1
echo "== whole tiles"; PERF_SCRATCH=0 timeout -k 10 200 python -u tests/perf_conv_gpu.py $CASES 2>&1 | grep -v amdgpu.ids || exit 1
echo "== split tiles"; VAMD_NT8_STREAMK=1 timeout -k 10 200 python -u tests/perf_conv_gpu.py $CASES 2>&1 | grep -v amdgpu.ids || exit 1
