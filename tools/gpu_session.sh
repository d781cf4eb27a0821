#!/bin/bash
# usage (GPU box, repo root): tools/gpu_session.sh <tag>
# GPU tests, bench.py with one lane per walker (comparison), then the round profile of the default geometry
TAG=${1:-s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
python3 bench.py --lanes 1 --no-cpu-baseline > $OUT/bench_lanes1.json 2> $OUT/bench_lanes1.err
cat $OUT/bench_lanes1.json
tools/gpu_round_profile.sh $TAG
