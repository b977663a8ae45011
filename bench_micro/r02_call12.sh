#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_merge.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r02/t12.log 2>&1; tail -3 gpurun_out/r02/t12.log
for c in 2 5; do timeout -k 10 600 python bench.py --config $c --no-scan --no-cpu-baseline > gpurun_out/r02/b12_c$c.json 2> gpurun_out/r02/b12_c$c.err; python -c "
import json; j=json.load(open('gpurun_out/r02/b12_c$c.json')); print('config $c', round(j['ms_per_step'],5), j['roofline']['kernel_ms'], j['verified']['ok'])"; done
timeout -k 10 600 python bench.py --force-sharded --no-scan --no-cpu-baseline > gpurun_out/r02/b12_sh.json 2> gpurun_out/r02/b12_sh.err; tail -2 gpurun_out/r02/b12_sh.err; python -c "
import json; j=json.load(open('gpurun_out/r02/b12_sh.json')); print('sharded world1', round(j['ms_per_step'],5), j['roofline']['kernel_ms'], j.get('verified'), j.get('host_enqueue_ms_per_step'))"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof12 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --force-sharded --no-scan --no-cpu-baseline --no-verify > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02/prof12.err
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r02/prof12 -name "*kernel_stats.csv" | head -1); echo $f; head -14 "$f"
