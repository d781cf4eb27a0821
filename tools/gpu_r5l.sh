#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
DET_WALKER=loco3d DET_MULTI_ONLY=1 timeout 1500 python3 tools/diag_determinism.py 128 40 > $O/det_loco3d.txt 2>&1; grep -c identical $O/det_loco3d.txt; grep -v identical $O/det_loco3d.txt | grep -v amdgpu | tail -5 | cut -c1-200
DET_WALKER=loco3d timeout 900 python3 tools/diag_determinism.py 24 4 > $O/det_loco3d_forms.txt 2>&1; grep -v amdgpu $O/det_loco3d_forms.txt | tail -12 | cut -c1-200
