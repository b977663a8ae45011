#!/bin/bash
# Round 2: HBM traffic of the config-3 scan kernels at 100M rows (int32 and int64 columns): two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE), then profiles/make_traffic_scan.py. Run on the GPU box from the repo root.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/traffic_scan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows 100000000"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pass_fetch -- python3 $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pass_write -- python3 $CMD > /dev/null 2> $OUT/write.err
cd $GRAFT_REPO_ROOT && python3 profiles/make_traffic_scan.py $OUT gpurun_out/r02/traffic_scan.json
