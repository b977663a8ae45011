#!/bin/bash
OUT=gpurun_out/r05/z; mkdir -p $OUT
timeout -k 10 500 python3 bench_micro/k1_by_allocation.py > $OUT/k1_by_allocation_1.log 2>&1; r=$?; grep -v amdgpu.ids $OUT/k1_by_allocation_1.log | tail -20
[ $r -eq 124 ] && exit 124
exit $r
