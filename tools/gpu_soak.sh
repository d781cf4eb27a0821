#!/bin/bash
# usage (GPU box): tools/gpu_soak.sh [passes] -- the -m gpu suite N times in fresh processes; prints every pass's verdict and the failures (flaky-failure watch)
cd $GRAFT_REPO_ROOT
out=gpurun_out/soak; mkdir -p $out
for i in $(seq 1 ${1:-10}); do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pass$i.log 2>&1
  echo "pass $i rc=$? $(tail -1 $out/pass$i.log)"
  grep -E "^(FAILED|ERROR)|Memory access fault|Aborted|core dumped" $out/pass$i.log | head -5
done
