#!/bin/bash
OUT=gpurun_out/r05/soak2; mkdir -p $OUT
export BMX_VIEW_DEBUG=1
for cfg in "3000000 150 int32 1" "1000000 200 int32 3" "6000000 80 int32 4"; do set -- $cfg
  timeout -k 10 400 python3 bench_micro/view_soak.py $1 $2 $3 $4 > $OUT/soak_$1_$3_$4.log 2>&1; r=$?
  grep -v "amdgpu.ids\|rewrite enqueued" $OUT/soak_$1_$3_$4.log | grep -v "^round" | tail -6
  if [ $r -ne 0 ]; then echo "rc=$r"; exit $r; fi
done
exit 0
