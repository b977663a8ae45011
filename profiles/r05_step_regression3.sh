#!/bin/bash
OUT=gpurun_out/r05/f; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-scan --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err; r=$?
  echo "$name rc=$r $(python -c "import json; j=json.load(open('$OUT/$name.json')); print('us/step %.2f event %.2f host %.2f' % (j['ms_per_step']*1e3, j['event_ms_per_step']*1e3, j.get('host_enqueue_ms_per_step',0)*1e3), j['roofline']['kernel_ms'], j.get('table_placement'))" 2>&1 | tail -1)"; if [ $r -eq 124 ]; then exit 124; fi; }
EXTRA="" run default A=1
EXTRA="" run no_touch BMX_NO_NULL_STREAM_TOUCH=1
EXTRA="" run skip_selfcheck BMX_SKIP_SELFCHECK=1
EXTRA="" run tries8 BMX_TABLE_PLACEMENT_TRIES=8
EXTRA="--no-defer" run nodefer A=1
EXTRA="" run default2 A=1
exit 0
