#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libdrloco_hip_prof.so timeout 600 python3 tools/diag_rollout_floor.py > $O/rollout_floor.txt 2>&1; grep -v amdgpu.ids $O/rollout_floor.txt | cut -c1-360
timeout 600 python3 tools/diag_randomize.py > $O/randomize_diag.txt 2>&1; grep -v amdgpu.ids $O/randomize_diag.txt
R=$GRAFT_REPO_ROOT/gpurun_out/r05_randomize; mkdir -p $R
python3 bench.py --no-cpu-baseline --randomize > $R/bench.json 2> $R/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --randomize > $R/bench_trace.log 2>&1
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --randomize --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/pmc1 -- $B > $R/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/pmc2 -- $B > $R/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_profile.py gpurun_out/r05_randomize gpurun_out/r05_sum r05_randomize randomize | sed -n '1,4p;16,40p' | cut -c1-200
