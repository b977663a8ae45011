# Round 5 soak (last code state): long verified streams with the deferred compaction on (run on the GPU box from the repo root); one line per run -> gpurun_out/r05/soak.log
set -o pipefail
mkdir -p gpurun_out/r05
L=gpurun_out/r05/soak.log
: > $L
run() { echo "== $*" >> $L; python3 bench.py "$@" 2>> gpurun_out/r05/soak.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=d.get('verified') or {}
print(json.dumps({'steps':d['steps'],'warmup':d['warmup'],'us_per_step':round(d['ms_per_step']*1000,2),'verified':v.get('ok') if isinstance(v,dict) else v,'batches_checked':v.get('batches') if isinstance(v,dict) else None,'winner_indices_compared':v.get('winner_indices_compared') if isinstance(v,dict) else None,'rows':v.get('rows') if isinstance(v,dict) else None,'deferred':(json.load(open('bench_detail.json')).get('deferred_compaction') if True else None),'exchange':(d.get('exchange') or {}).get('kind')}))" >> $L; }
run --config 5 --steps 590 --warmup 10 --no-scan --no-cpu-baseline && echo "soak 1 done" && \
run --config 2 --steps 300 --warmup 3 --no-scan --no-cpu-baseline && echo "soak 2 done" && \
run --config 5 --steps 230 --warmup 10 --no-scan --no-cpu-baseline --force-sharded && echo "soak 3 done" && \
cat $L
