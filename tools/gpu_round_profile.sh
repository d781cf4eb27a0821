#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round_profile.sh <round-tag> [walker]
# 1. bench.py default run (for the walker) -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same bench command -> kernel_stats csv
# 3. PMC passes (each its own run, --kernel-trace only next to --pmc) on tools/prof_step.py, which issues the benchmark's launch schedule
TAG=${1:-r02}
WALKER=${2:-straight}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
BW=""; if [ "$WALKER" != "straight" ]; then BW="--walker $WALKER"; fi
python3 bench.py $BW > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $BW > $OUT/bench_trace.log 2>&1
P="python3 $GRAFT_REPO_ROOT/tools/prof_step.py --walker $WALKER --steps 448 --warm 64"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc1 -- $P > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_IFETCH --kernel-trace --output-format csv -d $OUT/pmc2 -- $P > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -- $P > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc4 -- $P > $OUT/pmc4.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc5 -- $P > $OUT/pmc5.log 2>&1
# round 6: the DYNAMIC instruction mix: the VALU instruction classes the hardware counts, and the lanes that were active
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc6 -- $P > $OUT/pmc6.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d $OUT/pmc7 -- $P > $OUT/pmc7.log 2>&1
cat $OUT/bench.json
