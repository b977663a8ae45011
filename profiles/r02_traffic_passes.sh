#!/bin/bash
# Round 2: HBM traffic per k_probe_apply launch (final code): three separate rocprofv3 --pmc passes, then profiles/make_traffic.py.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pass_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pass_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pass_req -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify > /dev/null 2> $OUT/req.err
cd $GRAFT_REPO_ROOT && python3 profiles/make_traffic.py $OUT gpurun_out/r02/traffic_probe_apply.json
