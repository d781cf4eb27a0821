#!/bin/bash
TAG=${1:-r02e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "vecnormalize or overlap or steps_fixed or rollout or full_size or config0 or policy" > $OUT/pytest_focus.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_focus.log
tail -8 $OUT/pytest_focus.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
run default
run default_b
run policy --policy
run loco3d --walker loco3d
run randomize --randomize
