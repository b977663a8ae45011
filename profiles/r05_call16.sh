#!/bin/bash
OUT=gpurun_out/r05/y; mkdir -p $OUT
for cfg in "20 3" "20 30" "20 100" "100 3" "100 50" "20 3" "300 3"; do set -- $cfg
  timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-scan --no-cpu-baseline > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.err; r=$?
  python3 - $OUT/b_$1_$2.json $1 $2 <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps %s warmup %s: us/step %.2f event %.2f K1(events, own pass) %.2f placement %s" % (sys.argv[2], sys.argv[3], j["ms_per_step"]*1e3, j["event_ms_per_step"]*1e3, j["roofline"]["kernel_ms"]["probe_apply"]*1e3, j["table_placement"]))
PY
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
