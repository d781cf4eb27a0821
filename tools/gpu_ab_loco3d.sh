#!/bin/bash
# usage: tools/gpu_ab_loco3d.sh <variant .so under build_variants/>: bench.py --walker loco3d, product vs variant, alternating, three passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_loco3d; mkdir -p $O
run() { python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --walker loco3d 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,3), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  run product
  DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$1 run variant
done
