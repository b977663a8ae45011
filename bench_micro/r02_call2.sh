#!/bin/bash
set -e
mkdir -p gpurun_out/r02
./bench_micro/wb_micro > gpurun_out/r02/wb_micro.log 2>&1
cat gpurun_out/r02/wb_micro.log
rocprofv3 -L > gpurun_out/r02/counters_list.txt 2>&1 || true
grep -c . gpurun_out/r02/counters_list.txt
