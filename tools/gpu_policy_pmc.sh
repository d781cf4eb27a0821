#!/bin/bash
# usage (GPU box, repo root): tools/gpu_policy_pmc.sh <tag>
# The two persistent rollout kernels under rocprofv3: kernel trace + stats of `bench.py --policy [--moments per_rollout]`, then PMC passes (each its own run,
# --kernel-trace only next to --pmc) on the same command with --steps 2 --warmup 1.  Condensed by tools/summarize_profile.py into
# gpurun_out/<tag>_sum/<tag>_policy[_per_rollout]_summary.txt + traffic_policy[_per_rollout].json (copy to profiles/).
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
KINDS=${2:-"policy policy_per_rollout"}
for kind in $KINDS; do
  FLAGS="--policy"; [ $kind = policy_per_rollout ] && FLAGS="--policy --moments per_rollout"
  [ $kind = policy_walking ] && FLAGS="--policy --checkpoint walking --warmup 8"
  [ $kind = policy_walking_per_rollout ] && FLAGS="--policy --checkpoint walking --warmup 8 --moments per_rollout"
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_$kind; mkdir -p $OUT
  cd $GRAFT_REPO_ROOT
  python3 bench.py --no-cpu-baseline $FLAGS > $OUT/bench.json 2> $OUT/bench.err
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $FLAGS > $OUT/bench_trace.log 2>&1
  B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $FLAGS --steps 2"; case $kind in *walking*) ;; *) B="$B --warmup 1";; esac
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_IFETCH SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -- $B > $OUT/pmc3.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc4 -- $B > $OUT/pmc4.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc5 -- $B > $OUT/pmc5.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc6 -- $B > $OUT/pmc6.log 2>&1
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE --kernel-trace --output-format csv -d $OUT/pmc7 -- $B > $OUT/pmc7.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 tools/summarize_profile.py gpurun_out/${TAG}_$kind gpurun_out/${TAG}_sum ${TAG}_$kind $kind | sed -n '1,4p;16,60p' | cut -c1-220
done
