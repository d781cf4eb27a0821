#!/bin/bash
# usage: tools/gpu_ab_multi.sh <variant1.so> [<variant2.so> ...]: headline and --walker loco3d, product vs the variants under build_variants/, two alternating passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_multi; mkdir -p $O
run() { tag=$1; shift; python3 bench.py --no-cpu-baseline --steps 6 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,3), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/ab.txt; }
for i in 1 2; do
  run "product            straight"; run "product            loco3d  " --walker loco3d
  for v in "$@"; do
    DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$v run "$v straight"
    DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$v run "$v loco3d  " --walker loco3d
  done
done
