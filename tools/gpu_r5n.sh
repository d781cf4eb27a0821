#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
DET_WALKER=loco3d DET_MULTI_ONLY=1 timeout 1200 python3 tools/diag_determinism.py 128 200 > $O/det_loco3d.txt 2>&1; echo "19-dof split: identical repeats $(grep -c identical $O/det_loco3d.txt) of 199; differing: $(grep -c -i differ $O/det_loco3d.txt)"
DET_MULTI_ONLY=1 timeout 1200 python3 tools/diag_determinism.py 256 200 > $O/det_straight.txt 2>&1; echo "straight split: identical repeats $(grep -c identical $O/det_straight.txt) of 199; differing: $(grep -c -i differ $O/det_straight.txt)"
timeout 900 python3 tools/soak_persistent.py 2>&1 | grep -v amdgpu | cut -c1-200
DET_PERSISTENT=4096 timeout 900 python3 tools/diag_determinism_pairs.py 2>&1 | grep -v amdgpu | tail -3
