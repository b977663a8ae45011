#!/bin/bash
OUT=gpurun_out/r05/p; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_ordered_view.py tests/test_gpu_index_maintenance.py tests/test_gpu_scan.py -m gpu -q -x > $OUT/pytest_sel.log 2>&1; rc=$?; tail -6 $OUT/pytest_sel.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
[ $rc -ne 0 ] && exit $rc
for cfg in "10000000 int32" "100000000 int32" "100000000 wide"; do set -- $cfg
  timeout -k 10 300 python3 bench_micro/view_patch.py $1 $2 8 > $OUT/vp_$1_$2.log 2>&1; r=$?
  grep -v "amdgpu.ids\|^E2026\|^W2026" $OUT/vp_$1_$2.log | tail -5
  if [ $r -eq 124 ]; then exit 124; fi
done
timeout -k 10 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_scan.json 2> $OUT/bench_scan.err; r=$?; echo "bench rc=$r"; tail -c 1500 $OUT/bench_scan.json; [ -f bench_detail.json ] && cp bench_detail.json $OUT/bench_scan_detail.json
exit $r
