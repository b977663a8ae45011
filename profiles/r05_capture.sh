#!/bin/bash
# Round 5: evidence captures (on the GPU box from the repo root: bash profiles/r05_capture.sh <part>...); summaries are copied to profiles/ by hand.
#  line     the un-profiled default command: the short stdout line + bench_detail.json
#  stats    rocprofv3 --kernel-trace --stats of the default bench command (config 2) and of --config 5 (no scans, no CPU legs)
#  traffic  HBM traffic + request counts per k_probe_apply launch: three separate --pmc passes -> profiles/make_traffic.py
#  scans    rocprofv3 --kernel-trace --stats of the scan / view kernels, ONE index size per run: the program directly behind `--`
#  tscan    HBM traffic of the scan kernels at 100M rows: two --pmc passes -> profiles/make_traffic_scan.py
#  sharded  the N>1 code path on one GPU: world 1 --force-sharded (direct and RCCL-refused) with kernel stats, and the 2-rank rehearsal through the launcher
# A step that runs into its time limit ends the script (nothing is started on the GPU after it).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/final
mkdir -p $OUT/traffic $OUT/traffic_scan
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local r=$?; if [ $r -eq 124 ] || [ $r -eq 137 ]; then echo "TIME LIMIT: $*"; exit 124; fi; return $r; }
for part in "$@"; do
case $part in
line)
  (cd $GRAFT_REPO_ROOT && step 700 python3 bench.py > $OUT/n1_default_run.json 2> $OUT/n1_default.err; echo "rc=$? bytes=$(wc -c < $OUT/n1_default_run.json)"; cp bench_detail.json $OUT/n1_default_detail.json; cat $OUT/n1_default_run.json) ;;
stats)
  step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config2 -- python3 $B --no-cpu-baseline --no-scan > $OUT/n1_config2_run.json 2> $OUT/n1_config2.err
  step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config5 -- python3 $B --config 5 --no-scan --no-cpu-baseline > $OUT/n1_config5_run.json 2> $OUT/n1_config5.err
  for d in n1_config2 n1_config5; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/${d}_kernel_stats.csv; echo "== $d"; head -8 "$f" | cut -c1-170; done ;;
traffic)
  T="--steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify"
  step 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/traffic/pass_fetch -- python3 $B $T > /dev/null 2> $OUT/traffic/fetch.err
  step 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/traffic/pass_write -- python3 $B $T > /dev/null 2> $OUT/traffic/write.err
  step 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/traffic/pass_req -- python3 $B $T > /dev/null 2> $OUT/traffic/req.err
  (cd $GRAFT_REPO_ROOT && python3 profiles/make_traffic.py $OUT/traffic gpurun_out/r05/final/traffic_probe_apply.json | tail -n 20) ;;
scans)
  for rows in 10000000 100000000; do
    step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_$rows -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows $rows > $OUT/scan_${rows}_run.json 2> $OUT/scan_$rows.err
    f=$(find $OUT/scan_$rows -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/scan_${rows}_kernel_stats.csv; echo "== scans at $rows rows"; grep -E "k_scan|k_view|k_ordered|k_ix_update|Name" "$f" | cut -c1-200
  done ;;
tscan)
  S="--steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows 100000000"
  step 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/traffic_scan/pass_fetch -- python3 $B $S > /dev/null 2> $OUT/traffic_scan/fetch.err
  step 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/traffic_scan/pass_write -- python3 $B $S > /dev/null 2> $OUT/traffic_scan/write.err
  (cd $GRAFT_REPO_ROOT && python3 profiles/make_traffic_scan.py $OUT/traffic_scan gpurun_out/r05/final/traffic_scan.json | tail -n 5) ;;
sharded)
  step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sharded_w1_direct -- python3 $B --force-sharded --no-cpu-baseline --no-scan > $OUT/sharded_w1_direct_run.json 2> $OUT/sharded_w1_direct.err
  f=$(find $OUT/sharded_w1_direct -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/sharded_w1_direct_kernel_stats.csv; head -12 "$f" | cut -c1-170
  BMX_SHARDED_EXCHANGE=rccl step 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sharded_w1_rccl -- python3 $B --force-sharded --no-cpu-baseline --no-scan > $OUT/sharded_w1_rccl_run.json 2> $OUT/sharded_w1_rccl.err
  f=$(find $OUT/sharded_w1_rccl -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/sharded_w1_rccl_kernel_stats.csv; head -8 "$f" | cut -c1-170
  (cd $GRAFT_REPO_ROOT && BMX_BENCH_ONE_GPU_REHEARSAL=1 step 500 python3 bench.py --gpus 2 --no-scan --no-cpu-baseline > $OUT/rehearsal_2rank_run.json 2> $OUT/rehearsal_2rank.err; echo "rehearsal rc=$?"; cat $OUT/rehearsal_2rank_run.json | cut -c1-1500) ;;
esac
done
exit 0
