#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent.py tests/test_gpu_bench_shapes.py -m gpu -x -q -k "split or single_step_f32 or G4 or G3 or f32_randomization or rollout_f32 or persistent or bench or fixed or timeout" > gpurun_out/r4g/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4g/tests.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/r4g/$name.json 2> gpurun_out/r4g/$name.err; python3 -c "import json; d=json.load(open('gpurun_out/r4g/$name.json')); print('$name', round(d['value']/1e6,2), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_us'],1))" || tail -3 gpurun_out/r4g/$name.err; }
run default
run policy --policy
run policy_per_rollout --policy --moments per_rollout
run policy_launches --policy --rollout-form launches
