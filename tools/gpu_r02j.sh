#!/bin/bash
TAG=${1:-r02j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "vecnormalize or overlap or steps_fixed or rollout or group" > $OUT/pytest_focus.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_focus.log
tail -4 $OUT/pytest_focus.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
run default
run policy --policy
run policy_8192 --policy --envs-per-gpu 8192
run policy_8192_h2 --policy --envs-per-gpu 8192 --handles 2
run policy_16384 --policy --envs-per-gpu 16384 --steps 2
run policy_32768 --policy --envs-per-gpu 32768 --steps 2
run policy_32768_h2 --policy --envs-per-gpu 32768 --handles 2 --steps 2
run policy_65536_h4 --policy --envs-per-gpu 65536 --handles 4 --steps 1
