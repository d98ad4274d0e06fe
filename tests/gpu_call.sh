mkdir -p gpurun_out/c4
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/c4/gpu_tests.log 2>&1; rc=$?
tail -8 gpurun_out/c4/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/c4/bench_butterfly.json 2> gpurun_out/c4/bench_butterfly.err || { tail -20 gpurun_out/c4/bench_butterfly.err; exit 1; }
cat gpurun_out/c4/bench_butterfly.json
timeout -k 10 300 python bench.py --preset flowers --no-cpu-baseline > gpurun_out/c4/bench_flowers.json 2> gpurun_out/c4/bench_flowers.err || { tail -20 gpurun_out/c4/bench_flowers.err; exit 1; }
cat gpurun_out/c4/bench_flowers.json
timeout -k 10 400 python bench.py --preset ffhq --no-cpu-baseline --steps 10 --warmup 3 --sample-steps 50 > gpurun_out/c4/bench_ffhq.json 2> gpurun_out/c4/bench_ffhq.err || { tail -30 gpurun_out/c4/bench_ffhq.err; exit 1; }
cat gpurun_out/c4/bench_ffhq.json
timeout -k 10 120 python tests/diag_host_gpu.py flowers 16 20 > gpurun_out/c4/host_flowers.txt 2>&1 || { tail -20 gpurun_out/c4/host_flowers.txt; exit 1; }
head -3 gpurun_out/c4/host_flowers.txt
