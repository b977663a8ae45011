#!/bin/bash
# Same-box A/B of the JS host without the store: round 3's js/ (git archive d3663a7, untracked copy under bench_micro/ab/js_r3/, on the CURRENT addon and library)
# against the current js/; e2e_rate.js's default sections (mergeEntries synchronous, pipelined, typed columns), arms alternating, three rounds.
# Prepare here:  mkdir -p bench_micro/ab/js_r3 && git archive d3663a7 bullet-js_amd/js | tar -x -C bench_micro/ab/js_r3 --strip-components=1 && cp bullet-js_amd/bmx.node bullet-js_amd/libbmx.so bench_micro/ab/js_r3/
for r in 1 2 3; do
  for arm in r3 r4; do
    if [ $arm = r3 ]; then T=bench_micro/ab/js_r3/js; else T=bullet-js_amd/js; fi
    node $T/test/e2e_rate.js 1000000 500000 8 2>&1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$arm', 'sync %.2f M/s' % (d['mergeEntries_per_s']/1e6), 'pipelined %.2f M/s' % (d['mergeEntriesPipelined_per_s']/1e6), 'typed columns %.0f M/s' % (d['mergeBatch_typed_columns_per_s']/1e6))"
  done
done
