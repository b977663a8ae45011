#!/bin/bash
# same-box A/B of the scans: previous build (bench_micro/ab/prev/libbmx_prev.so: default-policy loads everywhere) against the tree's (nontemporal loads for value
# columns above 256 MiB); microseconds per scan, every timed scan verified inside bench.py
P=$GRAFT_REPO_ROOT/bench_micro/ab/prev/libbmx_prev.so
for r in 1 2; do for v in prev new; do
  if [ $v = prev ]; then export BMX_LIB_PATH=$P; else unset BMX_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=j['scan_config3']
def g(k,q,*p):
    d=sc[k][q]
    for x in p: d=d[x]
    return d
out=['$v']
for k in ('10M','10M_int64','100M','100M_int64'):
    out.append('| %s: count-only %.1f equals %.1f range10%% ids %.1f pos %.1f range50%% pos %.1f' % (k, g(k,'range_10pct','roofline_mask_kernel','count_only_scan_us'), g(k,'equals_0.1pct','us'), g(k,'range_10pct','us'), g(k,'range_10pct','position_output','us'), g(k,'range_50pct','position_output','us')))
print(' '.join(out))"
done; done
