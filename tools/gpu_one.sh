#!/bin/bash
# usage: tools/gpu_one.sh "<pytest -k expression>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/one
timeout 1500 python3 -m pytest tests -m gpu -x -q -s -k "$1" > gpurun_out/one/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/one/pytest.log
tail -15 gpurun_out/one/pytest.log
