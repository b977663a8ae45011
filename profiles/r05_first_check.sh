#!/bin/bash
# round 5, first GPU contact of the round: the GPU suite, the default bench line (short form), and the self-launched 2-rank rehearsal on one GPU
set -o pipefail
OUT=gpurun_out/r05/a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?; echo "bench rc=$rc bytes=$(wc -c < $OUT/bench_default.json)"; cp bench_detail.json $OUT/bench_default_detail.json 2>/dev/null
[ $rc -eq 0 ] || { tail -20 $OUT/bench_default.err; exit $rc; }
cat $OUT/bench_default.json
BMX_BENCH_ONE_GPU_REHEARSAL=1 python bench.py --gpus 2 --steps 6 --warmup 2 > $OUT/ranks2.json 2> $OUT/ranks2.err; rc=$?; echo "2 ranks rc=$rc"; cat $OUT/ranks2.json; grep "bench\[rank 0\]" $OUT/ranks2.err | tail -12
exit $rc
