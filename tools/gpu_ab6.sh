#!/bin/bash
# usage: tools/gpu_ab6.sh <variant.so under build_variants/> ...: the product's one-state build against experiment builds (built with -DDL_DPP_WAIT=1): headline, --walker loco3d and --policy,
# three alternating passes on one box, and a bit comparison of the float32 rollouts (tools/ab_bits.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab6; mkdir -p $O; : > $O/ab.txt
P=$GRAFT_REPO_ROOT/drloco_amd/csrc/libdrloco_hip_dpp1.so
run() { tag=$1; lib=$2; shift 2; DL_LIB_PATH=$lib python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,3), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/ab.txt; }
for i in 1 2 3; do
  run "product            straight" $P; run "product            loco3d  " $P --walker loco3d; run "product            policy  " $P --policy
  for v in "$@"; do
    run "$v straight" $GRAFT_REPO_ROOT/build_variants/$v; run "$v loco3d  " $GRAFT_REPO_ROOT/build_variants/$v --walker loco3d; run "$v policy  " $GRAFT_REPO_ROOT/build_variants/$v --policy
  done
done
DL_LIB_PATH=$P python3 tools/ab_bits.py $O/bits_product.npz > /dev/null 2>&1
for v in "$@"; do DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/$v python3 tools/ab_bits.py $O/bits_$v.npz > /dev/null 2>&1; echo "bits product vs $v:"; python3 tools/ab_bits.py --compare $O/bits_product.npz $O/bits_$v.npz | grep -c "identical=True"; python3 tools/ab_bits.py --compare $O/bits_product.npz $O/bits_$v.npz | grep "identical=False" | head -3; done
