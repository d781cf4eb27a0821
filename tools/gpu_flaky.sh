#!/bin/bash
# usage (GPU box): tools/gpu_flaky.sh [N]  -- the bit-identity test of the benchmark's launch shapes N times with the product build and with every build under build_variants/
cd $GRAFT_REPO_ROOT; shopt -s nullglob
N=${1:-12}
for lib in product build_variants/*.so; do
  [ "$lib" = product ] && unset DL_LIB_PATH || export DL_LIB_PATH=$GRAFT_REPO_ROOT/$lib
  fails=0
  for i in $(seq $N); do timeout 600 python3 -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -x -k "benchmark_launch_shapes or beyond_the_launch_cap" > /tmp/flaky.log 2>&1 || { fails=$((fails+1)); grep -m2 "^FAILED\|Error" /tmp/flaky.log; }; done
  echo "$lib: $fails failures in $N runs"
done
