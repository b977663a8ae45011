#!/bin/bash
OUT=gpurun_out/r05/t; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_ordered_view.py -m gpu -q -x > $OUT/pytest_sel.log 2>&1; rc=$?; tail -6 $OUT/pytest_sel.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 124
[ $rc -ne 0 ] && exit $rc
# A/B of the first sort of a view: rocPRIM's radix sort (default) against the patch path's own kernels over the whole column
for arm in library own; do
  if [ $arm = own ]; then export BMX_VIEW_SORT=own; else unset BMX_VIEW_SORT; fi
  for cfg in "10000000 int32" "100000000 int32" "100000000 wide"; do set -- $cfg
    timeout -k 10 300 python3 bench_micro/view_patch.py $1 $2 1 > $OUT/sort_${arm}_$1_$2.log 2>&1; r=$?
    echo "$arm: $(grep 'first sort' $OUT/sort_${arm}_$1_$2.log)"
    if [ $r -eq 124 ]; then exit 124; fi
  done
done
exit 0
