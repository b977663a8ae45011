#!/bin/bash
OUT=gpurun_out/r05/s; mkdir -p $OUT
timeout -k 10 200 ./bench_micro/view_merge_micro 100000000 > $OUT/view_merge_micro.log 2>&1; r=$?; tail -22 $OUT/view_merge_micro.log; [ $r -eq 124 ] && exit 124
timeout -k 10 600 python -m pytest tests/test_gpu_ordered_view.py tests/test_gpu_index_maintenance.py tests/test_gpu_scan.py -m gpu -q -x > $OUT/pytest_sel.log 2>&1; rc=$?; tail -6 $OUT/pytest_sel.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
[ $rc -ne 0 ] && exit $rc
for cfg in "10000000 int32" "100000000 int32" "100000000 wide"; do set -- $cfg
  timeout -k 10 300 python3 bench_micro/view_patch.py $1 $2 9 > $OUT/vp_$1_$2.log 2>&1; r=$?
  grep -v "amdgpu.ids\|^E2026\|^W2026" $OUT/vp_$1_$2.log | tail -5
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
