#!/bin/bash
# Round 3: the N>1 code path on the one-GPU box (bash profiles/r03_capture_sharded.sh from the repo root).
#  1. one-rank rehearsal (bench.py --force-sharded, process group of one): direct exchange vs the RCCL slab exchange, un-profiled lines, config 2 and config 5
#  2. rocprofv3 --kernel-trace --stats of the direct rehearsal
#  3. two and four ranks sharing the GPU (gloo control plane, IPC data plane: BMX_BENCH_ONE_GPU_REHEARSAL=1): bench.py's own N>1 logic incl. its verification
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/sharded
mkdir -p $OUT
B=$GRAFT_REPO_ROOT/bench.py
cd $GRAFT_REPO_ROOT
for x in direct rccl; do
  BMX_SHARDED_EXCHANGE=$x python3 $B --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_${x}_plain_run.json 2> $OUT/world1_$x.err
  BMX_SHARDED_EXCHANGE=$x python3 $B --config 5 --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_${x}_config5_plain_run.json 2> $OUT/world1_${x}_c5.err
done
cd /tmp && export TMPDIR=/tmp
BMX_SHARDED_EXCHANGE=direct rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/world1_direct -- python3 $B --force-sharded --no-scan --no-cpu-baseline > $OUT/world1_direct_run.json 2> $OUT/world1_direct_prof.err
f=$(find $OUT/world1_direct -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/world1_direct_kernel_stats.csv; head -12 "$f" | cut -c1-170
cd $GRAFT_REPO_ROOT
for n in 2 4; do
  BMX_BENCH_ONE_GPU_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2956$n bench.py --gpus $n --steps 10 --warmup 3 > $OUT/ranks${n}_one_gpu_rehearsal.json 2> $OUT/ranks$n.err
  tail -c 900 $OUT/ranks${n}_one_gpu_rehearsal.json; echo
done
BMX_BENCH_ONE_GPU_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29570 bench.py --gpus 4 --config 5 --steps 12 --warmup 4 > $OUT/ranks4_config5_one_gpu_rehearsal.json 2> $OUT/ranks4_c5.err
tail -c 600 $OUT/ranks4_config5_one_gpu_rehearsal.json; echo
for f in $OUT/world1_*_plain_run.json; do python3 -c "
import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['ms_per_step']*1e3,1), j['exchange']['kind'], j['verified'] and j['verified']['ok'])"; done
