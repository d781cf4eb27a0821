#!/bin/bash
TAG=${1:-r02i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 600 python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python3 -c "import json; d=json.load(open('$OUT/bench.json')); print(round(d['value']/1e6,2))"
