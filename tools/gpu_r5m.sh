#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
DET_WALKER=loco3d DET_MULTI_ONLY=1 timeout 2400 python3 tools/diag_determinism.py 128 400 > $O/det_loco3d.txt 2>&1; echo "19-dof split: identical repeats $(grep -c identical $O/det_loco3d.txt) of 399; differing: $(grep -c -i differ $O/det_loco3d.txt)"
DET_MULTI_ONLY=1 timeout 2400 python3 tools/diag_determinism.py 256 300 > $O/det_straight.txt 2>&1; echo "straight split: identical repeats $(grep -c identical $O/det_straight.txt) of 299; differing: $(grep -c -i differ $O/det_straight.txt)"
