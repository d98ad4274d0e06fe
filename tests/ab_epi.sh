# A/B of the ping-pong kernel against the 4-wave kernels with epilogue terms switched on (PERF_EPI bit mask)
CASES="64,112,112,128,128,3,1,0 64,112,112,64,128,3,1,0 64,112,112,64,128,1,1,0 64,56,56,128,256,3,1,0"
for E in 1 2 6; do for V in 1 0; do echo "== PERF_EPI=$E VAMD_NT8=$V"; PERF_EPI=$E VAMD_NT8=$V PERF_SCRATCH=0 timeout -k 10 100 python -u tests/perf_conv_gpu.py $CASES 2>&1 | grep -v amdgpu.ids | cut -c1-150 || exit 1; done; done
