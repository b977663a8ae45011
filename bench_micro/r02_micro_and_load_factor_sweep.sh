#!/bin/bash
# round 2, GPU call 1: no-atomic two-pass microbenchmark + load-factor sweep of the round-1 merge kernel with line buckets
set -e
mkdir -p gpurun_out/r02
hipcc -O3 --offload-arch=gfx950 -o bench_micro/twopass_micro bench_micro/twopass_micro.hip && ./bench_micro/twopass_micro > gpurun_out/r02/twopass_micro.log 2>&1
cat gpurun_out/r02/twopass_micro.log
for lp in 35 50 60 70 80 90; do
  BMX_BENCH_CAP=15000000 BMX_BENCH_LOAD_PCT=$lp timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02/lf_$lp.json 2> gpurun_out/r02/lf_$lp.err
  python - <<PY
import json
j=json.load(open("gpurun_out/r02/lf_$lp.json"))
print("load_pct $lp", "ms/step", round(j["ms_per_step"],5), "kernel_ms", j["roofline"]["kernel_ms"], "uniq", j["unique_keys_mode"]["kernel_ms"])
PY
done
