#!/bin/bash
OUT=gpurun_out/r05/w; mkdir -p $OUT

rc=0
[ $rc -eq 124 ] && exit 124
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for cfg in "10000000 int32" "100000000 int32" "100000000 wide"; do set -- $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/vp_$1_$2 -- python3 $GRAFT_REPO_ROOT/bench_micro/view_patch.py $1 $2 9 > $GRAFT_REPO_ROOT/$OUT/vp_$1_$2.log 2>&1; r=$?
  grep -v "amdgpu.ids\|^E2026\|^W2026" $GRAFT_REPO_ROOT/$OUT/vp_$1_$2.log | tail -5
  f=$(find $GRAFT_REPO_ROOT/$OUT/vp_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$OUT/vp_$1_$2_kernel_stats.csv
  python3 - $GRAFT_REPO_ROOT/$OUT/vp_$1_$2_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_view','k_ix_update','PredLogCreated','PredChanged','PredFlag','PredNotIn')):
        print("  %-44s calls %3s avg %9.1f us" % (n.split('(')[0].replace('void bmx::','')[:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
  if [ $r -eq 124 ]; then exit 124; fi
done
exit 0
