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
