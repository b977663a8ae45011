#!/bin/bash
# A/B of library builds on ONE box (build each variant to bench_micro/ab/libbmx_<name>.so; BMX_LIB_PATH selects it): bench.py config 2 at 10 % and 0 % inserts and config 5, each variant twice, interleaved.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  for p in 10 0; do
    BMX_LIB_PATH=$GRAFT_REPO_ROOT/bench_micro/ab/libbmx_$v.so BMX_BENCH_INSERT_PCT=$p python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v ins$p', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'], j['unique_keys_mode']['kernel_ms']['probe_apply'])"
  done
  BMX_LIB_PATH=$GRAFT_REPO_ROOT/bench_micro/ab/libbmx_$v.so python bench.py --config 5 --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v config5', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done
done
