#!/bin/bash
OUT=gpurun_out/r05/d; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; r=$?
  echo "$name rc=$r $(python -c "import json; j=json.load(open('$OUT/$name.json')); print('us/step %.2f event %.2f host %.2f' % (j['ms_per_step']*1e3, j['event_ms_per_step']*1e3, j.get('host_enqueue_ms_per_step',0)*1e3), j['roofline']['kernel_ms'])" 2>&1 | tail -1)"; if [ $r -eq 124 ]; then exit 124; fi; }
EXTRA="" run default A=1
EXTRA="" run null_stream BMX_SELFCHECK_NULL_STREAM=1
EXTRA="" run skip BMX_SKIP_SELFCHECK=1
for v in "A=1" "BMX_SKIP_SELFCHECK=1"; do
  ( cd bench_micro/ab/r5_bisect/9e8fd8f && env $v timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/tree_r04_$v.json 2> $GRAFT_REPO_ROOT/$OUT/tree_r04_$v.err ); r=$?
  echo "tree r04 $v rc=$r $(python -c "import json; j=json.loads(open('$OUT/tree_r04_$v.json').read().strip().splitlines()[-1]); print('us/step %.2f event %.2f' % (j['ms_per_step']*1e3, j['event_ms_per_step']*1e3), j['roofline']['kernel_ms'], j.get('deferred_compaction'))" 2>&1 | tail -1)"
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
