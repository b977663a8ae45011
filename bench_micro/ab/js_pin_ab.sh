cd bullet-js_amd/js
nproc; lscpu | grep -E "NUMA|Socket|Model name" | head -6
for r in 1 2 3; do
  echo "plain:   $(node test/e2e_rate.js 1000000 200000 5 only apply 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["applied_path"]["batchSync_apply_entries_per_s"]))')"
  echo "taskset: $(taskset -c 2 node test/e2e_rate.js 1000000 200000 5 only apply 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["applied_path"]["batchSync_apply_entries_per_s"]))')"
done
