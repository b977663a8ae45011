#!/bin/bash
# round 5, second GPU call: the GPU suite (all failures listed), the default bench line, the self-launched 2-rank rehearsal, and the sharded one-rank A/B
# (compaction in-stream / deferred on a third stream / deferred on the exchange stream)
set -o pipefail
OUT=gpurun_out/r05/b; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -8 $OUT/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
timeout -k 10 420 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc2=$?; echo "bench rc=$rc2 bytes=$(wc -c < $OUT/bench_default.json)"; cp bench_detail.json $OUT/bench_default_detail.json 2>/dev/null
[ $rc2 -eq 124 ] && exit 124
[ $rc2 -eq 0 ] || tail -20 $OUT/bench_default.err
cat $OUT/bench_default.json
BMX_BENCH_ONE_GPU_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 6 --warmup 2 > $OUT/ranks2.json 2> $OUT/ranks2.err; rc3=$?; echo "2 ranks rc=$rc3"; cat $OUT/ranks2.json; grep "bench\[rank 0\]" $OUT/ranks2.err | tail -12
[ $rc3 -eq 124 ] && exit 124
for rep in 1 2; do for d in 0 1 2; do
  BMX_SHARDED_DEFER=$d timeout -k 10 200 python bench.py --force-sharded --no-cpu-baseline > $OUT/sharded_defer${d}_$rep.json 2> $OUT/sharded_defer${d}_$rep.err; r=$?
  echo "defer=$d rep=$rep rc=$r $(python -c "import json,sys; j=json.load(open('$OUT/sharded_defer${d}_$rep.json')); print('us/step %.2f' % (j['ms_per_step']*1e3), j['roofline']['kernel_ms'], j['verified'])" 2>&1 | tail -1)"
  [ $r -eq 124 ] && exit 124
done; done
exit $(( rc + rc2 + rc3 ))
