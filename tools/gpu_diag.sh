#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/diag
python3 tools/diag_sections.py > gpurun_out/diag/sections_straight.txt 2>&1; cat gpurun_out/diag/sections_straight.txt | grep -v amdgpu.ids
python3 tools/diag_sections.py --walker loco3d > gpurun_out/diag/sections_loco3d.txt 2>&1; cat gpurun_out/diag/sections_loco3d.txt | grep -v amdgpu.ids
