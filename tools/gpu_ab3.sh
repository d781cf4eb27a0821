#!/bin/bash
# usage (GPU box): tools/gpu_ab3.sh <tag> -- product vs every build under build_variants/, three lines each (headline, --policy, --policy --moments per_rollout), two alternating passes
TAG=${1:-ab3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT; shopt -s nullglob
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step')" || tail -3 $OUT/$name.err; }
for pass in 1 2; do
  for lib in product build_variants/*.so; do
    if [ "$lib" = product ]; then unset DL_LIB_PATH; v=product; else export DL_LIB_PATH=$GRAFT_REPO_ROOT/$lib; v=$(basename $lib .so); v=${v#libdrloco_hip_}; fi
    run ${v}_$pass
    run ${v}_policy_$pass --policy
    run ${v}_perrollout_$pass --policy --moments per_rollout
  done
done
