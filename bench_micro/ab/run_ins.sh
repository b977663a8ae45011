#!/bin/bash
# timing-only variants at 10 % inserts (results of the variants are wrong on purpose: --no-verify), interleaved, three repetitions; "cur ins0" = the reference point
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  BMX_LIB_PATH=$GRAFT_REPO_ROOT/bench_micro/ab/libbmx_$v.so BMX_BENCH_INSERT_PCT=10 timeout -k 10 120 python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v ins10', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms']['probe_apply'])"
done
BMX_LIB_PATH=$GRAFT_REPO_ROOT/bench_micro/ab/libbmx_cur.so BMX_BENCH_INSERT_PCT=0 timeout -k 10 120 python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('cur ins0', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms']['probe_apply'])"
done
