#!/bin/bash
OUT=gpurun_out/r05/v; mkdir -p $OUT
for i in 1 2 3 4; do
  timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline > $OUT/b$i.json 2> $OUT/b$i.err; r=$?
  python3 - $OUT/b$i.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("us/step %.2f event %.2f K1 %.2f K2 %.2f K3 %.2f placement %s" % (j["ms_per_step"]*1e3, j["event_ms_per_step"]*1e3, j["roofline"]["kernel_ms"]["probe_apply"]*1e3, j["roofline"]["kernel_ms"]["resolve_lists"]*1e3, j["roofline"]["kernel_ms"]["compact"]*1e3, j["table_placement"]))
PY
  if [ $r -eq 124 ]; then exit 124; fi
done
BMX_TABLE_PLACEMENT_TRIES=8 timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline > $OUT/b8.json 2> $OUT/b8.err
python3 - $OUT/b8.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tries 8: us/step %.2f event %.2f K1 %.2f placement %s" % (j["ms_per_step"]*1e3, j["event_ms_per_step"]*1e3, j["roofline"]["kernel_ms"]["probe_apply"]*1e3, j["table_placement"]))
PY
rocm-smi --showclocks 2>/dev/null | head -20
exit 0
