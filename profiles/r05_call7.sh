#!/bin/bash
# Round 5: bench_micro/view_patch.py under rocprofv3 --kernel-trace --stats, one index shape per run (the program directly behind `--`); summaries -> profiles/r05_view_patch_*_kernel_stats.csv
OUT=gpurun_out/r05/w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "10000000 int32" "100000000 int32" "100000000 wide"; do set -- $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/vp_$1_$2 -- python3 $GRAFT_REPO_ROOT/bench_micro/view_patch.py $1 $2 9 > $GRAFT_REPO_ROOT/$OUT/vp_$1_$2.log 2>&1; r=$?
  grep -v "amdgpu.ids\|^E2026\|^W2026" $GRAFT_REPO_ROOT/$OUT/vp_$1_$2.log | tail -5
  f=$(find $GRAFT_REPO_ROOT/$OUT/vp_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$OUT/vp_$1_$2_kernel_stats.csv
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
