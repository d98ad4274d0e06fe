# One GPU-box call: the whole GPU suite, the allocation-free entry point's test, then the three bench presets.
# Every GPU step has its own time limit and the chain stops at the first failure.
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -15 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for f in tests/wip_no_alloc.py; do
  [ -f $f ] || continue
  timeout -k 10 200 python -u -m pytest $f -x -q --timeout 120 --timeout-method thread > gpurun_out/no_alloc.log 2>&1; rc=$?
  tail -5 gpurun_out/no_alloc.log
  [ $rc -eq 0 ] || exit $rc
done
timeout -k 10 300 python bench.py > gpurun_out/bench_butterfly.json 2> gpurun_out/bench_butterfly.err || { tail -20 gpurun_out/bench_butterfly.err; exit 1; }
cat gpurun_out/bench_butterfly.json
timeout -k 10 300 python bench.py --preset flowers --no-cpu-baseline > gpurun_out/bench_flowers.json 2> gpurun_out/bench_flowers.err || { tail -20 gpurun_out/bench_flowers.err; exit 1; }
cat gpurun_out/bench_flowers.json
VAMD_BENCH_LAYERS=1 timeout -k 10 400 python bench.py --preset ffhq --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/bench_ffhq.json 2> gpurun_out/bench_ffhq.err || { tail -30 gpurun_out/bench_ffhq.err; exit 1; }
cat gpurun_out/bench_ffhq.json; grep "^layer" gpurun_out/bench_ffhq.err | head -40
