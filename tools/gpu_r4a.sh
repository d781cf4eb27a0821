#!/bin/bash
# round 4, first GPU call: the new split-form parity tests, the lockstep experiment, a bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "G4 or G5 or G11_mirrored or divergence or evaluation_mode or monitor_lists or f32_randomization or f32_error_growth or split_workgroups" -s > gpurun_out/r4a/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4a/tests.log
timeout 900 python3 tools/diag_lockstep.py --save gpurun_out/r4a/lockstep.npz > gpurun_out/r4a/lockstep.log 2>&1
echo "lockstep rc=$?"; cat gpurun_out/r4a/lockstep.log | tail -30
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
echo "bench rc=$?"; python3 -c "import json; d=json.load(open('gpurun_out/r4a/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4a/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/r4a/smoke.log
