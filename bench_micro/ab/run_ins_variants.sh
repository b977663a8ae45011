#!/bin/bash
# Round-3 A/B on ONE box: where k_probe_apply's absent keys get their rows (BMX_K1_INSERTS=inline|block|launch, merge_kernels.h), each variant
# on config 2 at 10 % and 0 % inserts and on config 5, twice, interleaved; plus (when bench_micro/ab/r1_tree exists: `git archive aaeb569`
# built in place, not tracked) the complete round-1 tree on the same box for VERDICT r2 weak #6.
# usage: bash bench_micro/ab/run_ins_variants.sh [variants...]    output: one line per run on stdout
cd $GRAFT_REPO_ROOT
VARS="${@:-inline block launch}"
for rep in 1 2; do
for v in $VARS; do
  for p in 10 0; do
    BMX_K1_INSERTS=$v BMX_BENCH_INSERT_PCT=$p python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v ins$p', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'], 'unique', j['unique_keys_mode']['kernel_ms']['probe_apply'])"
  done
  BMX_K1_INSERTS=$v python bench.py --config 5 --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v config5', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done
if [ -f bench_micro/ab/r1_tree/bench.py ]; then
  for p in 10 0; do
    (cd bench_micro/ab/r1_tree && BMX_BENCH_INSERT_PCT=$p python bench.py --no-cpu-baseline 2>/dev/null) | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('round1-tree ins$p', round(j['ms_per_step']*1e3,1), j['roofline'].get('kernel_ms'))"
  done
fi
done
