#!/bin/bash
# round 5, third GPU call: (1) bisect the unsharded step regression; (2) ordered-view tests + JS parity; (3) kernel trace of the view patch at 10M / 100M rows
OUT=gpurun_out/r05/c; mkdir -p $OUT
bash profiles/r05_step_regression.sh; [ $? -eq 124 ] && exit 124
timeout -k 10 500 python -m pytest tests/test_gpu_ordered_view.py tests/test_js_host.py tests/test_gpu_sharded_ranks.py -m gpu -q > $OUT/pytest_sel.log 2>&1; rc=$?; tail -6 $OUT/pytest_sel.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
cd /tmp && export TMPDIR=/tmp
for cfg in "10000000 int32" "100000000 int32" "100000000 wide"; do set -- $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/vp_$1_$2 -- python3 $GRAFT_REPO_ROOT/bench_micro/view_patch.py $1 $2 3 > $GRAFT_REPO_ROOT/$OUT/vp_$1_$2.log 2>&1; r=$?
  cat $GRAFT_REPO_ROOT/$OUT/vp_$1_$2.log | grep -v amdgpu.ids | tail -5
  f=$(find $GRAFT_REPO_ROOT/$OUT/vp_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp $f $GRAFT_REPO_ROOT/$OUT/vp_$1_$2_kernel_stats.csv; grep -E "k_view|k_ix_update|k_sel|k_ordered|Name" $f | cut -c1-200; }
  [ $r -eq 124 ] && exit 124
done
exit 0
