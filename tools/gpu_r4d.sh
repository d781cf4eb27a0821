#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split or single_step_f32 or G4 or f32_randomization or f32_error_growth or rollout_f32" > gpurun_out/r4d/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4d/tests.log
for i in 1 2; do timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r4d/bench$i.json 2> gpurun_out/r4d/bench$i.err; python3 -c "import json; d=json.load(open('gpurun_out/r4d/bench$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" || tail -5 gpurun_out/r4d/bench$i.err; done
for v in prof1 prof2 prof3; do echo $v; DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libdrloco_hip_$v.so timeout 300 python3 tools/diag_split.py 2>&1 | tail -1; done
