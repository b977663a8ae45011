#!/bin/bash
OUT=gpurun_out/r05/q; mkdir -p $OUT
for cfg in "100000000 int32"; do set -- $cfg
  BMX_VIEW_DEBUG=1 timeout -k 10 300 python3 bench_micro/view_patch.py $1 $2 9 > $OUT/vp_$1_$2.log 2>&1; r=$?
  grep -v "amdgpu.ids\|^E2026\|^W2026" $OUT/vp_$1_$2.log | tail -12
  if [ $r -eq 124 ]; then exit 124; fi
done
