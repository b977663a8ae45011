set -o pipefail
mkdir -p gpurun_out/r04/v2
python -m pytest tests -m gpu -x -q > gpurun_out/r04/v2/t_all.log 2>&1 || { tail -30 gpurun_out/r04/v2/t_all.log; exit 1; }
tail -3 gpurun_out/r04/v2/t_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04/v2/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --force-sharded --no-cpu-baseline > gpurun_out/r04/v2/sharded_w1.json 2> gpurun_out/r04/v2/sharded_w1.err; echo "sharded rc=$?"
python bench.py --force-sharded --no-cpu-baseline --config 5 > gpurun_out/r04/v2/sharded_w1_c5.json 2> gpurun_out/r04/v2/sharded_w1_c5.err; echo "sharded c5 rc=$?"
BMX_BENCH_ONE_GPU_REHEARSAL=1 python3 bench.py --gpus 4 --steps 6 --warmup 2 > gpurun_out/r04/v2/ranks4.json 2> gpurun_out/r04/v2/ranks4.err; echo "4 ranks rc=$?"
(cd bullet-js_amd/js && node test/device_parity.js > ../../gpurun_out/r04/v2/device_parity.log 2>&1; echo "device_parity rc=$?")
