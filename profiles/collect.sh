#!/bin/bash
# Runs on the GPU box (gpurun -- bash profiles/collect.sh): the TCC counter passes first (each in a run of its own,
# counters only) and the traffic file stamped with the kernel sources' hash, then the bench lines of the five presets
# (which read that file), the SQ counter pass, and the kernel trace + statistics of the training bench of butterfly,
# flowers and ffhq (profiles/trace_presets.sh).  python profiles/summarize.py condenses the result afterwards.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/collect
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$ROOT'); from vision_amd import _lib; print(_lib.source_hash())" > $OUT/source_hash.txt
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-sampling --one-stream"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $BENCH --steps 1 --warmup 1 > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $BENCH --steps 1 --warmup 1 > $OUT/write.log 2>&1
echo "write done"
(cd $ROOT && python profiles/summarize.py --traffic-only) && echo "traffic file written"
(cd $ROOT && timeout -k 10 300 python bench.py > $OUT/bench_butterfly.json 2> $OUT/bench_butterfly.err) && echo "butterfly done"
(cd $ROOT && timeout -k 10 300 python bench.py --preset flowers --no-cpu-baseline > $OUT/bench_flowers.json 2> $OUT/bench_flowers.err) && echo "flowers done"
(cd $ROOT && timeout -k 10 400 python bench.py --preset ffhq --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_ffhq.json 2> $OUT/bench_ffhq.err) && echo "ffhq done"
(cd $ROOT && timeout -k 10 300 python bench.py --preset vqvaeunet > $OUT/bench_vqvaeunet.json 2> $OUT/bench_vqvaeunet.err) && echo "first stage done"
(cd $ROOT && timeout -k 10 300 python bench.py --preset ldm > $OUT/bench_ldm.json 2> $OUT/bench_ldm.err) && echo "ldm done"
(cd $ROOT && timeout -k 10 300 python bench.py --graph --no-cpu-baseline --no-sampling > $OUT/bench_butterfly_graph.json 2> $OUT/bench_butterfly_graph.err) && echo "graph replay done"
(cd $ROOT && timeout -k 10 300 python bench.py --one-stream --no-cpu-baseline --no-sampling > $OUT/bench_butterfly_one_stream.json 2> $OUT/bench_butterfly_one_stream.err) && echo "one stream done"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- $BENCH --steps 1 --warmup 1 > $OUT/sq.log 2>&1
echo "sq done"
bash $ROOT/profiles/trace_presets.sh
