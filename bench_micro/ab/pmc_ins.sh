#!/bin/bash
# PMC comparison of k_probe_apply: batches without absent keys vs batches whose absent keys are probed but not created (timing-only build)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_ins
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name lib insert_pct
  export BMX_LIB_PATH=$GRAFT_REPO_ROOT/bench_micro/ab/libbmx_$2.so BMX_BENCH_INSERT_PCT=$3
  i=0
  for line in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT/$1_pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify > /dev/null 2> $OUT/$1_pass$i.err || echo "failed $1 $i"
  done
}
run ins0 cur 0
run absent_probed_only nocreate 10
run ins10 cur 10
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r02/pmc_ins")
for name in ("ins0", "absent_probed_only", "ins10"):
    acc = {}
    for f in glob.glob(os.path.join(root, name + "_pass*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_probe_apply<false, 0, false>" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(name, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
