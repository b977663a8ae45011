#!/bin/bash
mkdir -p gpurun_out/r02
for ph in 1 2 3 0; do
  BMX_DEBUG_PHASE=$ph timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02/b4_ph$ph.json 2> gpurun_out/r02/b4_ph$ph.err
  python - <<PY
import json
j=json.load(open("gpurun_out/r02/b4_ph$ph.json"))
print("phase $ph", "ms/step", round(j["ms_per_step"],5), "kernel_ms", j["roofline"]["kernel_ms"])
PY
done
