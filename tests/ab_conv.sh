# side-by-side microbenchmark of two builds of the library on the same GPU: ab_conv.sh <libB> <cases...>
LIBB=$1; shift
for rep in 1 2; do
  echo "== A (default)"; timeout -k 10 200 python -u tests/perf_conv_gpu.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
  echo "== B ($LIBB)"; VAMD_LIB=$LIBB timeout -k 10 200 python -u tests/perf_conv_gpu.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
