#!/bin/bash
# usage (GPU box): tools/gpu_ab.sh <build_variants/lib.so> -- bit comparison of the float32 rollouts of the product build with another build; then (second argument given) a soak run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
python3 tools/ab_bits.py gpurun_out/ab/new.npz
DL_LIB_PATH=$GRAFT_REPO_ROOT/$1 python3 tools/ab_bits.py gpurun_out/ab/base.npz
python3 tools/ab_bits.py --compare gpurun_out/ab/base.npz gpurun_out/ab/new.npz
rm -f gpurun_out/ab/*.npz
if [ -n "$2" ]; then timeout 900 python3 tools/soak.py 2>&1 | tail -12; fi
