# does keeping node on ONE NUMA node's cores steady the store-kept seam? (2 x EPYC 9575F: node 0 = CPUs 0-63,128-191); arms alternating
cd bullet-js_amd/js
rate() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["applied_path"]["batchSync_apply_entries_per_s"]))'; }
for r in 1 2 3 4; do
  echo "plain        : $(node test/e2e_rate.js 1000000 200000 5 only apply 2>/dev/null | rate)"
  echo "cpus 0-31    : $(taskset -c 0-31 node test/e2e_rate.js 1000000 200000 5 only apply 2>/dev/null | rate)"
  echo "cpus 64-95   : $(taskset -c 64-95 node test/e2e_rate.js 1000000 200000 5 only apply 2>/dev/null | rate)"
done
