#!/bin/bash
# usage (GPU box, repo root): tools/gpu_traffic.sh <tag>  -- HBM traffic of the env-step kernel (separate FETCH_SIZE / WRITE_SIZE passes)
TAG=${1:-traffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --steps 448 --warm 64 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc4 -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py --steps 448 --warm 64 > $OUT/pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_profile.py $OUT /tmp/traffic_sum $TAG | grep -E "FETCH_SIZE|WRITE_SIZE|hbm_bytes"
