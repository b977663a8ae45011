#!/bin/bash
# Round-3 A/B, third pass (one box, interleaved, two repetitions): variants given as arguments (BMX_K1_INSERTS values), config 2 at 10 % / 0 % inserts and config 5.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  for p in 10 0; do
    BMX_K1_INSERTS=$v BMX_BENCH_INSERT_PCT=$p python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v ins$p', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'], 'unique', j['unique_keys_mode']['kernel_ms']['probe_apply'])"
  done
  BMX_K1_INSERTS=$v python bench.py --config 5 --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v config5', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done
done
