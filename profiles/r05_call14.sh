#!/bin/bash
OUT=gpurun_out/r05/x; mkdir -p $OUT
timeout -k 10 400 python3 bench_micro/k1_input_placement.py 4 48 > $OUT/k1_input_placement.log 2>&1; r=$?; grep -v "amdgpu.ids" $OUT/k1_input_placement.log | tail -12
exit $r
