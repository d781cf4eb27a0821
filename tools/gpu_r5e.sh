#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -q -s -k "config5_full_size or error_growth" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "cumulative fraction|passed|failed|rc=|Error|assert" $O/pytest.log | cut -c1-400
DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libdrloco_hip_prof.so timeout 600 python3 tools/diag_rollout_floor.py > $O/rollout_floor.txt 2>&1; cat $O/rollout_floor.txt | cut -c1-330
tools/gpu_policy_pmc.sh r05
