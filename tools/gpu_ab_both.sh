#!/bin/bash
# usage: tools/gpu_ab_both.sh <variant .so under build_variants/>: headline and --walker loco3d, product vs variant, alternating, three passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_both; mkdir -p $O
run() { tag=$1; shift; python3 bench.py --no-cpu-baseline --steps 6 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,3), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  run "product  straight"
  DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$1 run "variant  straight"
  run "product  loco3d  " --walker loco3d
  DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$1 run "variant  loco3d  " --walker loco3d
done
