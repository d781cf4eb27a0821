#!/bin/bash
# usage: tools/gpu_rest.sh <tag> "<-k expression>"     part of the -m gpu suite, no -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-rest}; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -q -k "$2" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
