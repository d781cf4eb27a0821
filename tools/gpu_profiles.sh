#!/bin/bash
# usage (GPU box, repo root): tools/gpu_profiles.sh <round tag, e.g. r03>  -- the round's profiles:
#   <tag>, <tag>_loco3d                  both walkers: bench.py, rocprofv3 kernel trace + stats of the same command, PMC passes (gpu_round_profile.sh)
#   <tag>_policy[_per_rollout|_launches] the policy in the loop: persistent kernel with exact per-step moments (bench.py --policy), with per-rollout
#                                        moments, and the launch-per-step form; kernel trace + stats each, one PMC pass on the persistent kernel
# and condenses each directory into profiles-ready files under gpurun_out/<tag>_sum/ (tools/summarize_profile.py; copy them to profiles/).
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
tools/gpu_round_profile.sh $TAG straight > /dev/null
tools/gpu_round_profile.sh ${TAG}_loco3d loco3d > /dev/null
pol() {   # name, extra bench flags
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_$1; shift
  mkdir -p $OUT
  cd $GRAFT_REPO_ROOT
  python3 bench.py --no-cpu-baseline --policy "$@" > $OUT/bench.json 2> $OUT/bench.err
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy "$@" > $OUT/bench_trace.log 2>&1
  cd $GRAFT_REPO_ROOT
}
pol policy
pol policy_per_rollout --moments per_rollout
pol policy_launches --rollout-form launches
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_policy
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy --steps 2 --warmup 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_IFETCH --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy --steps 2 --warmup 1 > $OUT/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/diag_sections.py > gpurun_out/${TAG}_sum_sections_straight.txt 2>&1
S=gpurun_out/${TAG}_sum
python3 tools/summarize_profile.py gpurun_out/$TAG $S $TAG straight | tail -3
python3 tools/summarize_profile.py gpurun_out/${TAG}_loco3d $S ${TAG}_loco3d loco3d | tail -3
python3 tools/summarize_profile.py gpurun_out/${TAG}_policy $S ${TAG}_policy policy | head -14
python3 tools/summarize_profile.py gpurun_out/${TAG}_policy_per_rollout $S ${TAG}_policy_per_rollout policy_per_rollout | head -8
python3 tools/summarize_profile.py gpurun_out/${TAG}_policy_launches $S ${TAG}_policy_launches policy_launches | head -10
