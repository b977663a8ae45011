#!/bin/bash
# Round 5: the whole GPU suite + the JS host's device tests on one box (bash profiles/r05_full_tests.sh)
OUT=gpurun_out/r05/full; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 240 node bullet-js_amd/js/test/device_parity.js > $OUT/js_device_parity.log 2>&1; rc=$?; tail -4 $OUT/js_device_parity.log; echo "js rc=$rc"
exit $rc
