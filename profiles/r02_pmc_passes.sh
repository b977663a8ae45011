#!/bin/bash
# Round 2: stall attribution of k_probe_apply (config 2). Separate rocprofv3 --pmc passes (<= 4 TCC counters each) over a short bench run;
# profiles/r02_pmc_summarize.py turns the CSVs into profiles/r02_pmc_probe_apply.json. Run on the GPU box from the repo root.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed: $line"
  echo "pass $i done: $line"
done <<'PASSES'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_BUSY_sum TCC_CYCLE_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
TCC_IB_STALL_sum TCC_IB_REQ_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TA_TCP_STATE_READ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
PASSES
cd $GRAFT_REPO_ROOT && python3 profiles/r02_pmc_summarize.py $OUT gpurun_out/r02/r02_pmc_probe_apply.json
