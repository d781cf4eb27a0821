#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_dynamics or single_step or rollout_f64 or loco3d_rollout or loco3d_single or loco3d_forward or split or randomization" > gpurun_out/r4h/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4h/tests.log
bash tools/gpu_variants.sh r4h_ab 2>&1 | tail -8
