#!/bin/bash
# Round-3 A/B, second pass (one box, interleaved, two repetitions): barrier-free K1 variants and measurement-only switches (BMX_K1_DBG) that
# take single costs of a row creation out of the inline kernel (results are wrong with them: --no-verify, never shipped).
cd $GRAFT_REPO_ROOT
run() { # label, env...
  local label=$1; shift
  for p in 10 0; do
    env "$@" BMX_BENCH_INSERT_PCT=$p python bench.py --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$label ins$p', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'], 'unique', j['unique_keys_mode']['kernel_ms']['probe_apply'])"
  done
}
for rep in 1 2; do
  run inline BMX_K1_INSERTS=inline
  run nobar BMX_K1_INSERTS=nobar
  run lastwave BMX_K1_INSERTS=lastwave
  run inline+plainpublish BMX_K1_INSERTS=inline BMX_K1_DBG=1
  run inline+noctr BMX_K1_INSERTS=inline BMX_K1_DBG=2
  run inline+nocreatestore BMX_K1_INSERTS=inline BMX_K1_DBG=4
  for v in nobar lastwave; do
    BMX_K1_INSERTS=$v python bench.py --config 5 --no-scan --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v config5', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
  done
done
