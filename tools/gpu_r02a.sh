#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r02a.sh <tag>  -- tests of the 16-lane kernels incl. the 19-dof walker, then the two bench lines
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q -k "loco3d or forward or row_primitives or randomization" -s > $OUT/pytest_focus.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_focus.log
tail -15 $OUT/pytest_focus.log
timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench_straight.json 2> $OUT/bench_straight.err; cat $OUT/bench_straight.json
timeout 600 python3 bench.py --no-cpu-baseline --walker loco3d > $OUT/bench_loco3d.json 2> $OUT/bench_loco3d.err; cat $OUT/bench_loco3d.json; tail -3 $OUT/bench_loco3d.err
timeout 600 python3 bench.py --no-cpu-baseline --walker loco3d --lanes 1 --steps 1 > $OUT/bench_loco3d_l1.json 2> $OUT/bench_loco3d_l1.err; cat $OUT/bench_loco3d_l1.json
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
