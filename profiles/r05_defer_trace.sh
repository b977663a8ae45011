#!/bin/bash
# round 5: kernel trace of the timed region with the deferral fast (the null stream was used before the context's streams were created) and slow (it was not)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  BMX_SELFCHECK_NULL_STREAM=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-scan --no-cpu-baseline --no-verify --steps 12 --warmup 3 > $OUT/tr$v.json 2> $OUT/tr$v.err; r=$?
  f=$(find $OUT/tr$v -name "*kernel_trace.csv" | head -1)
  echo "== BMX_SELFCHECK_NULL_STREAM=$v rc=$r"; python3 -c "import json; j=json.load(open('$OUT/tr$v.json')); print('us/step %.2f' % (j['ms_per_step']*1e3))"
  python3 $GRAFT_REPO_ROOT/bench_micro/trace_timeline.py $f 30 3 | tee $OUT/timeline$v.txt
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
