#!/bin/bash
# usage (GPU box): tools/gpu_walking_restamp.sh <tag> -- the walking-policy parts of tools/gpu_profiles_r6.sh and tools/gpu_table.sh again (after bench.py / diag_walking.py started
# every rollout from the checkpoint's moments): PMC passes of both persistent kernels, diag_walking on the product and the -DDL_EXP_ROLLOUT_PROF=2 build, the table's walking lines
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
S=gpurun_out/${TAG}_sum; mkdir -p $S
tools/gpu_policy_pmc.sh $TAG "policy_walking policy_walking_per_rollout" > /dev/null
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/diag_walking.py 6 2>&1 | grep -v amdgpu.ids > $S/${TAG}_walking_summary.txt
if [ -f build_variants/libdrloco_hip_prof.so ]; then
  echo "" >> $S/${TAG}_walking_summary.txt; echo "---- the same with the -DDL_EXP_ROLLOUT_PROF=2 build (per-step phase records of the exact mode; the records cost a few per cent):" >> $S/${TAG}_walking_summary.txt
  DL_LIB_PATH=$PWD/build_variants/libdrloco_hip_prof.so timeout 900 python3 tools/diag_walking.py 4 2>&1 | grep -v amdgpu.ids >> $S/${TAG}_walking_summary.txt
fi
cp $S/traffic_env_step_policy_walking*.json profiles/ 2>/dev/null
OUT=gpurun_out/${TAG}table_walking; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch', d['roofline'].get('from_profile'))" || tail -3 $OUT/$name.err; }
run walking --policy --checkpoint walking --warmup 8 --steps 10
run walking_per_rollout --policy --checkpoint walking --warmup 8 --steps 10 --moments per_rollout
run walking_launches --policy --checkpoint walking --warmup 8 --steps 6 --rollout-form launches
run walking_deterministic --policy --checkpoint walking --warmup 8 --steps 10 --deterministic
run walking_32768 --policy --checkpoint walking --warmup 2 --steps 4 --envs-per-gpu 32768
run walking_long --policy --checkpoint walking --warmup 8 --steps 150
