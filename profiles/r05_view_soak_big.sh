#!/bin/bash
# Round 5: the soak of the view maintenance at 10^8 rows (pending patch, background rewrite, swaps), every answer against numpy
OUT=gpurun_out/r05/soak_big; mkdir -p $OUT
for cfg in "100000000 40 int32 7" "60000000 40 wide 8"; do set -- $cfg
  timeout -k 10 520 python3 bench_micro/view_soak.py $1 $2 $3 $4 > $OUT/soak_$1_$3_$4.log 2>&1; r=$?
  grep -v "amdgpu.ids" $OUT/soak_$1_$3_$4.log | tail -4
  if [ $r -ne 0 ]; then echo "rc=$r"; exit $r; fi
done
exit 0
