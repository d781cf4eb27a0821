#!/bin/bash
# usage (GPU box): tools/gpu_det.sh [R] -- R repetitions of the 512-step benchmark launch per build (product + build_variants/*): how many differ from the first
cd $GRAFT_REPO_ROOT; shopt -s nullglob
R=${1:-150}
for lib in product build_variants/*.so; do
  [ "$lib" = product ] && unset DL_LIB_PATH || export DL_LIB_PATH=$GRAFT_REPO_ROOT/$lib
  n=$(DET_MULTI_ONLY=1 timeout 1200 python3 tools/diag_determinism.py 512 $R 2>&1 | grep -c DIFFERS)
  echo "$lib: $n of $R runs differ"
done
