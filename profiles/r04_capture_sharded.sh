#!/bin/bash
# Round 4: the one-rank rehearsal of the process-per-GPU pipeline under rocprofv3 --kernel-trace --stats, the un-profiled lines (config 2, config 5; direct and RCCL exchange)
# (bash profiles/r04_capture_sharded.sh from the repo root on the GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/sharded
mkdir -p $OUT
B=$GRAFT_REPO_ROOT/bench.py
cd $GRAFT_REPO_ROOT
python3 $B --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_direct_plain_run.json 2> $OUT/world1_direct.err < /dev/null && echo "direct plain done" && \
python3 $B --config 5 --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_direct_config5_plain_run.json 2> $OUT/world1_direct_c5.err < /dev/null && echo "direct config 5 done" && \
BMX_SHARDED_EXCHANGE=rccl python3 $B --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_rccl_plain_run.json 2> $OUT/world1_rccl.err < /dev/null && echo "rccl plain done" && \
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/world1_direct -- python3 $B --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_direct_run.json 2> $OUT/world1_direct_prof.err < /dev/null && \
f=$(find $OUT/world1_direct -name "*kernel_stats.csv" | head -1) && [ -n "$f" ] && cp "$f" $OUT/world1_direct_kernel_stats.csv && head -12 "$f" | cut -c1-170
