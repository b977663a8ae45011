#!/bin/bash
# round 5: (7) id-output memory policies A/B, (3) sharded one-rank rehearsal A/B (wait folded into the scatter / deferral variants), (6) placement counters
OUT=gpurun_out/r05/j; mkdir -p $OUT
timeout -k 10 300 python bench_micro/scan_nt_emit_ab.py 100000000 > $OUT/scan_nt_emit_ab.log 2>&1; r=$?; grep -v amdgpu.ids $OUT/scan_nt_emit_ab.log; [ $r -eq 124 ] && exit 124
for rep in 1 2; do for cfg in "1 0" "0 0" "1 1" "1 2" "0 1"; do set -- $cfg
  BMX_PART_WAIT_FOLD=$1 BMX_SHARDED_DEFER=$2 timeout -k 10 200 python bench.py --force-sharded --no-cpu-baseline > $OUT/sh_fold$1_defer$2_$rep.json 2> $OUT/sh_fold$1_defer$2_$rep.err; r=$?
  echo "fold=$1 defer=$2 rep=$rep rc=$r $(python -c "import json; j=json.load(open('$OUT/sh_fold$1_defer$2_$rep.json')); print('us/step %.2f' % (j['ms_per_step']*1e3), j['roofline']['kernel_ms'], j['verified'])" 2>&1 | tail -1)"
  [ $r -eq 124 ] && exit 124
done; done
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; BMX_TABLE_PLACEMENT_TRIES=8 BMX_PLACEMENT_DEBUG=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pc_$name -- python3 $GRAFT_REPO_ROOT/bench_micro/placement_counters.py > $GRAFT_REPO_ROOT/$OUT/pc_$name.log 2>&1; r=$?
  echo "pmc pass $name rc=$r"; grep "bmx placement\|placement:" $GRAFT_REPO_ROOT/$OUT/pc_$name.log | tail -9; if [ $r -eq 124 ]; then exit 124; fi; }
pass utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum
pass stall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum
pass chan TCC_EA0_RDREQ
find $GRAFT_REPO_ROOT/$OUT -name "*.csv" | head -20
exit 0
