#!/bin/bash
# usage (GPU box, repo root): tools/gpu_variants.sh <tag>  -- bench every experiment build under build_variants/ next to the product build
TAG=${1:-variants}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
run product
run product_policy --policy
for f in build_variants/*.so; do
  v=$(basename $f .so); v=${v#libdrloco_hip_}
  export DL_LIB_PATH=$GRAFT_REPO_ROOT/$f
  run $v
  run ${v}_policy --policy
  unset DL_LIB_PATH
done
run product_again
