#!/bin/bash
mkdir -p gpurun_out/r02
( time timeout -k 10 900 python bench.py > gpurun_out/r02/b6_default.json 2> gpurun_out/r02/b6_default.err ) 2> gpurun_out/r02/b6_default.time
tail -3 gpurun_out/r02/b6_default.time; tail -3 gpurun_out/r02/b6_default.err
python - <<PY
import json
j=json.load(open("gpurun_out/r02/b6_default.json"))
print("config2 ms/step", round(j["ms_per_step"],5), "value", round(j["value"]/1e9,3), j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "verified", j["verified"])
for k,v in j["scan_config3"].items():
    print(k, {n:(x["us"], x["frac_of_8TBs"], x["roofline_mask_kernel"]["kernel_us"], x["roofline_mask_kernel"]["frac"]) for n,x in v.items() if isinstance(x,dict)}, v["index_build_ms"])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("all_cores",{}).get("value"), j["cpu_baseline"].get("js_twin"))
PY
for bk in 0 1; do
BMX_BENCH_BUCKETED=$bk timeout -k 10 600 python bench.py --config 5 --no-scan --no-cpu-baseline > gpurun_out/r02/b6_c5_bk$bk.json 2> gpurun_out/r02/b6_c5_bk$bk.err
tail -2 gpurun_out/r02/b6_c5_bk$bk.err
python - <<PY
import json
j=json.load(open("gpurun_out/r02/b6_c5_bk$bk.json"))
print("config5 bucketed=$bk ms/step", round(j["ms_per_step"],5), "value", round(j["value"]/1e9,3), j["roofline"]["kernel_ms"], "verified", j["verified"])
PY
done
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02/t6.log 2>&1; tail -4 gpurun_out/r02/t6.log
