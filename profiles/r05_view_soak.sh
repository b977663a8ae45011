#!/bin/bash
# Round 5: soak of the view maintenance (bench_micro/view_soak.py), several shapes; every answer checked against numpy
OUT=gpurun_out/r05/soak; mkdir -p $OUT
for cfg in "3000000 150 int32 1" "3000000 150 wide 2" "1000000 200 int32 3" "6000000 80 int32 4" "2000000 150 wide 5" "3000000 150 int32 6"; do set -- $cfg
  timeout -k 10 400 python3 bench_micro/view_soak.py $1 $2 $3 $4 > $OUT/soak_$1_$3_$4.log 2>&1; r=$?
  grep -v "amdgpu.ids" $OUT/soak_$1_$3_$4.log | tail -2
  if [ $r -ne 0 ]; then echo "rc=$r"; exit $r; fi
done
exit 0
