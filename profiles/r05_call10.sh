#!/bin/bash
# round 5: whole GPU suite + the default bench line + 2-rank rehearsal (normal and with an injected failure) on the current build
OUT=gpurun_out/r05/i; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -8 $OUT/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
timeout -k 10 420 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc2=$?; echo "bench rc=$rc2 bytes=$(wc -c < $OUT/bench_default.json)"; cp bench_detail.json $OUT/bench_default_detail.json 2>/dev/null
[ $rc2 -eq 124 ] && exit 124
[ $rc2 -eq 0 ] || grep -v "^bench detail" $OUT/bench_default.err | tail -20
cat $OUT/bench_default.json
exit $(( rc + rc2 ))
