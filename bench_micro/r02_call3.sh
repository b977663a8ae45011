#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_merge.py -m gpu -q -x > gpurun_out/r02/t3.log 2>&1 || true
tail -5 gpurun_out/r02/t3.log
for leg in 0 1; do
  BMX_BENCH_LEGACY=$leg timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02/b3_leg$leg.json 2> gpurun_out/r02/b3_leg$leg.err
  python - <<PY
import json
j=json.load(open("gpurun_out/r02/b3_leg$leg.json"))
print("legacy $leg", "ms/step", round(j["ms_per_step"],5), "value", round(j["value"]/1e9,3), "kernel_ms", j["roofline"]["kernel_ms"], "winners", j["winners_per_step"])
PY
done
for lp in 35 50 70; do
  BMX_BENCH_CAP=15000000 BMX_BENCH_LOAD_PCT=$lp timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02/b3_lf$lp.json 2> gpurun_out/r02/b3_lf$lp.err
  python - <<PY
import json
j=json.load(open("gpurun_out/r02/b3_lf$lp.json"))
print("new path cap15M load_pct $lp", "ms/step", round(j["ms_per_step"],5), "kernel_ms", j["roofline"]["kernel_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02/prof3.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02/prof3.err
cd $GRAFT_REPO_ROOT && find gpurun_out/r02/prof3 -name "*kernel_stats*" | head -3
f=$(find gpurun_out/r02/prof3 -name "*kernel_stats.csv" | head -1); head -12 "$f"
