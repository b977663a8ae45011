#!/bin/bash
# round 5: why does the unsharded step take 144 us (round 4: 77.7)? one variable at a time, same box
OUT=gpurun_out/r05/c; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; r=$?
  echo "$name rc=$r $(python -c "import json; j=json.load(open('$OUT/$name.json')); print('us/step %.2f event %.2f host %.2f' % (j['ms_per_step']*1e3, j['event_ms_per_step']*1e3, j.get('host_enqueue_ms_per_step',0)*1e3), j['roofline']['kernel_ms'], j.get('table_placement'))" 2>&1 | tail -1)"; if [ $r -eq 124 ]; then exit 124; fi; }
EXTRA="" run default A=1
EXTRA="--no-defer" run nodefer A=1



EXTRA="" run default2 A=1
# the same measurement from the trees of earlier commits (their own bench.py and libbmx.so), same box
for c in 9e8fd8f 8ea77e0; do
  ( cd bench_micro/ab/r5_bisect/$c && timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/tree_$c.json 2> $GRAFT_REPO_ROOT/$OUT/tree_$c.err ); r=$?
  echo "tree $c rc=$r $(python -c "import json; j=json.loads(open('$OUT/tree_$c.json').read().strip().splitlines()[-1]); print('us/step %.2f event %.2f' % (j['ms_per_step']*1e3, j['event_ms_per_step']*1e3), j['roofline']['kernel_ms'], j.get('deferred_compaction'))" 2>&1 | tail -1)"
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
