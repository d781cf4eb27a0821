#!/bin/bash
# usage (GPU box, repo root): tools/gpu_variants.sh <tag> [policy]  -- bench every experiment build under build_variants/ next to the product build
# (twice, in alternating order: the boxes drift by a few tenths of a percent within a call); "policy": the policy-in-the-loop line as well
TAG=${1:-variants}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
for pass in 1 2; do
  run product_$pass
  [ -n "$2" ] && run product_policy_$pass --policy
  for f in build_variants/*.so; do
    v=$(basename $f .so); v=${v#libdrloco_hip_}
    export DL_LIB_PATH=$GRAFT_REPO_ROOT/$f
    run ${v}_$pass
    [ -n "$2" ] && run ${v}_policy_$pass --policy
    unset DL_LIB_PATH
  done
done
run product_3
