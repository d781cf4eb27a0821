#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or single_step_f32 or G4 or f32_randomization or f32_error_growth or rollout_f32" > gpurun_out/r4e/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4e/tests.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/r4e/$name.json 2> gpurun_out/r4e/$name.err; python3 -c "import json; d=json.load(open('gpurun_out/r4e/$name.json')); print('$name', round(d['value']/1e6,2), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],1))" || tail -3 gpurun_out/r4e/$name.err; }
for i in 1 2; do
run product_$i

done

