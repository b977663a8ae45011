#!/bin/bash
# Round 2: rocprofv3 --kernel-trace --stats captures of the final code (run on the GPU box from the repo root); summaries are copied to profiles/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/n1_config2_run.json 2> $OUT/n1_config2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config5 -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --no-scan --no-cpu-baseline > $OUT/n1_config5_run.json 2> $OUT/n1_config5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sharded_world1 -- python3 $GRAFT_REPO_ROOT/bench.py --force-sharded --no-scan --no-cpu-baseline > $OUT/sharded_world1_run.json 2> $OUT/sharded_world1.err
BMX_BENCH_BUCKETED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config2_bucketed -- python3 $GRAFT_REPO_ROOT/bench.py --no-scan --no-cpu-baseline > $OUT/n1_config2_bucketed_run.json 2> $OUT/n1_config2_bucketed.err
cd $GRAFT_REPO_ROOT
for d in n1_config2 n1_config5 sharded_world1 n1_config2_bucketed; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/${d}_kernel_stats.csv; echo "== $d"; head -9 "$f" | cut -c1-160; done
python3 bench.py > $OUT/n1_default_run.json 2> $OUT/n1_default.err
python3 bench_micro/host_mode_rate.py > $OUT/host_mode_rate.log 2>&1; tail -5 $OUT/host_mode_rate.log
