#!/bin/bash
OUT=gpurun_out/r05/ab3; mkdir -p $OUT
for rep in 1 2; do
for defer in 0 2; do
for waves in 4 5 6 8; do
  BMX_SHARDED_DEFER=$defer BMX_K1_WAVES=$waves timeout -k 10 200 python bench.py --force-sharded --no-scan --no-cpu-baseline > $OUT/b_${defer}_${waves}_$rep.json 2> $OUT/b_${defer}_${waves}_$rep.err; r=$?
  python3 - $OUT/b_${defer}_${waves}_$rep.json $defer $waves $rep $r <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("defer=%s waves=%s rep=%s rc=%s us/step %.2f %s %s" % (sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], j["ms_per_step"]*1e3, j["roofline"]["kernel_ms"], j["verified"]))
except Exception as e:
    print("defer=%s waves=%s rep=%s rc=%s no line (%s)" % (sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], e))
PY
  if [ $r -eq 124 ]; then exit 124; fi
done; done; done
exit 0
