#!/bin/bash
OUT=gpurun_out/r05/z2; mkdir -p $OUT
timeout -k 10 500 python3 bench_micro/k1_by_allocation.py > $OUT/k1_by_allocation.log 2>&1; r=$?; grep -v amdgpu.ids $OUT/k1_by_allocation.log | tail -6
[ $r -ne 0 ] && exit $r
timeout -k 10 500 python3 bench_micro/step_over_time.py 8 > $OUT/step_over_time_a.log 2>&1; grep -v amdgpu.ids $OUT/step_over_time_a.log | tail -9
timeout -k 10 500 python3 bench_micro/step_over_time.py 8 > $OUT/step_over_time_b.log 2>&1; grep -v amdgpu.ids $OUT/step_over_time_b.log | tail -9
timeout -k 10 500 python3 bench_micro/step_over_time.py 8 > $OUT/step_over_time_c.log 2>&1; grep -v amdgpu.ids $OUT/step_over_time_c.log | tail -9
exit 0
