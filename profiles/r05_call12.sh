#!/bin/bash
# round 5: JS host with the opt-in lazy store: device parity (fixtures + stress + the lazy-store test), then the seam's rate eager / lazy
OUT=gpurun_out/r05/k; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_js_host.py -m gpu -q > $OUT/pytest_js.log 2>&1; rc=$?; tail -30 $OUT/pytest_js.log | cut -c1-400; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
cd bullet-js_amd/js
for key in apply lazy; do
  timeout -k 10 300 node test/e2e_rate.js 1000000 200000 5 only $key > ../../$OUT/e2e_$key.json 2> ../../$OUT/e2e_$key.err; r=$?; echo "e2e $key rc=$r"; cat ../../$OUT/e2e_$key.json | cut -c1-600; tail -3 ../../$OUT/e2e_$key.err
  [ $r -eq 124 ] && exit 124
done
exit $rc
