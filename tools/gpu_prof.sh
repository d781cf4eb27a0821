#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag>
# kernel trace + stats, then PMC passes (each its own run), all as CSV under gpurun_out/<tag>/
TAG=${1:-prof}
VAR=${2:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --variant $VAR > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --variant $VAR --steps 448 --warm 64 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_IFETCH --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --variant $VAR --steps 448 --warm 64 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --variant $VAR --steps 448 --warm 64 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc4 -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --variant $VAR --steps 448 --warm 64 > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
