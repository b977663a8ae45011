#!/bin/bash
OUT=gpurun_out/r05/u; mkdir -p $OUT
for arm in library own; do
  if [ $arm = own ]; then export BMX_VIEW_SORT=own; else unset BMX_VIEW_SORT; fi
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $OUT/bench_$arm.json 2> $OUT/bench_$arm.err; r=$?
  python3 - $OUT/bench_$arm.json $arm <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k:(v.get("view_sort_ms"), v.get("view_first_equals_after_merge_us_worst_of_4"), v.get("view_first_equals_after_merge_us_best_of_4")) for k,v in j["scan_config3"].items()})
PY
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
