#!/bin/bash
# table-size sweep on ONE box: capacity 15M rows, table = capacity*100/load_pct slots of 32 B; config 2 (10 % inserts: 10M -> 14.3M rows over the run)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lp in "$@"; do
  BMX_BENCH_CAP=15000000 BMX_BENCH_LOAD_PCT=$lp timeout -k 10 120 python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('load_pct $lp slots %.1fM' % (15e6*100/$lp/1e6), round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'], j['unique_keys_mode']['kernel_ms']['probe_apply'])"
done
done
