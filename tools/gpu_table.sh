#!/bin/bash
# usage (GPU box): tools/gpu_table.sh <tag> -- the benchmark line and its variants, one bench.py run each
TAG=${1:-table}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
run default
run policy --policy
run policy_per_rollout --policy --moments per_rollout
run policy_launches --policy --rollout-form launches
run randomize --randomize
run nosplit --no-split
run loco3d --walker loco3d
run envs8192 --envs-per-gpu 8192
run envs32768 --envs-per-gpu 32768
run policy_32768_h2 --policy --envs-per-gpu 32768 --handles 2
run policy_8192 --policy --envs-per-gpu 8192
run policy_16384 --policy --envs-per-gpu 16384
run policy_32768 --policy --envs-per-gpu 32768
run policy_32768_per_rollout --policy --envs-per-gpu 32768 --moments per_rollout
# round 6: the trained (walking) policy, the spec-conformant code object, the 19-dof walker with a policy
run walking --policy --checkpoint walking --warmup 8 --steps 10
run walking_per_rollout --policy --checkpoint walking --warmup 8 --steps 10 --moments per_rollout
run walking_launches --policy --checkpoint walking --warmup 8 --steps 6 --rollout-form launches
run walking_deterministic --policy --checkpoint walking --warmup 8 --steps 10 --deterministic
run walking_32768 --policy --checkpoint walking --warmup 2 --steps 4 --envs-per-gpu 32768
run loco3d_policy --walker loco3d --policy
run loco3d_policy_per_rollout --walker loco3d --policy --moments per_rollout
run policy_h256 --policy --hidden 256
run policy_h128 --policy --hidden 128
DL_DPP_WAIT=2 run default_two_wait_states
DL_DPP_WAIT=2 run loco3d_two_wait_states --walker loco3d
DL_DPP_WAIT=2 run policy_two_wait_states --policy
