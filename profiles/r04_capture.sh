#!/bin/bash
# Round 4: evidence captures (run on the GPU box from the repo root: bash profiles/r04_capture.sh <part>...); summaries are copied to profiles/ by hand.
#  stats    rocprofv3 --kernel-trace --stats of the default bench command (config 2) and of --config 5 (no scans, no CPU legs)
#  traffic  HBM traffic + request counts per k_probe_apply launch: three separate --pmc passes -> profiles/make_traffic.py
#  scans    rocprofv3 --kernel-trace --stats of the scan kernels, ONE index size per run (10M rows, 100M rows): the program directly behind `--`
#  tscan    HBM traffic of the scan kernels at 100M rows: two --pmc passes -> profiles/make_traffic_scan.py
#  line     the un-profiled default line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/final
mkdir -p $OUT/traffic $OUT/traffic_scan
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
for part in "$@"; do
case $part in
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config2 -- python3 $B --no-cpu-baseline --no-scan > $OUT/n1_config2_run.json 2> $OUT/n1_config2.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1_config5 -- python3 $B --config 5 --no-scan --no-cpu-baseline > $OUT/n1_config5_run.json 2> $OUT/n1_config5.err
  for d in n1_config2 n1_config5; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/${d}_kernel_stats.csv; echo "== $d"; head -8 "$f" | cut -c1-170; done ;;
traffic)
  T="--steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/traffic/pass_fetch -- python3 $B $T > /dev/null 2> $OUT/traffic/fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/traffic/pass_write -- python3 $B $T > /dev/null 2> $OUT/traffic/write.err
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/traffic/pass_req -- python3 $B $T > /dev/null 2> $OUT/traffic/req.err
  (cd $GRAFT_REPO_ROOT && python3 profiles/make_traffic.py $OUT/traffic gpurun_out/r04/final/traffic_probe_apply.json | tail -n 20) ;;
scans)
  for rows in 10000000 100000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scan_$rows -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows $rows > $OUT/scan_${rows}_run.json 2> $OUT/scan_$rows.err
    f=$(find $OUT/scan_$rows -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/scan_${rows}_kernel_stats.csv; echo "== scans at $rows rows"; grep -E "k_scan|Name" "$f" | cut -c1-200
  done ;;
tscan)
  S="--steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows 100000000"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/traffic_scan/pass_fetch -- python3 $B $S > /dev/null 2> $OUT/traffic_scan/fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/traffic_scan/pass_write -- python3 $B $S > /dev/null 2> $OUT/traffic_scan/write.err
  (cd $GRAFT_REPO_ROOT && python3 profiles/make_traffic_scan.py $OUT/traffic_scan gpurun_out/r04/final/traffic_scan.json | tail -n 5) ;;
line)
  (cd $GRAFT_REPO_ROOT && python3 bench.py > $OUT/n1_default_run.json 2> $OUT/n1_default.err; tail -c 600 $OUT/n1_default_run.json) ;;
esac
done
