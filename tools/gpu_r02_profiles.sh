#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r02_profiles.sh  -- the round's profiles: both walkers (kernel trace + PMC passes) and the policy-in-the-loop trace
cd $GRAFT_REPO_ROOT
tools/gpu_round_profile.sh r02 straight > /dev/null
tools/gpu_round_profile.sh r02_loco3d loco3d > /dev/null
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_policy
mkdir -p $OUT
python3 bench.py --no-cpu-baseline --policy > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy > $OUT/bench_trace.log 2>&1
OUT2=$GRAFT_REPO_ROOT/gpurun_out/r02_policy_32768_h2
mkdir -p $OUT2
cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline --policy --envs-per-gpu 32768 --handles 2 --steps 2 > $OUT2/bench.json 2> $OUT2/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT2/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy --envs-per-gpu 32768 --handles 2 --steps 2 > $OUT2/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/diag_sections.py > gpurun_out/r02_sum_sections_straight.txt 2>&1
python3 tools/diag_sections.py --walker loco3d > gpurun_out/r02_sum_sections_loco3d.txt 2>&1
python3 tools/summarize_profile.py gpurun_out/r02_policy_32768_h2 gpurun_out/r02_sum r02_policy_32768_h2 policy | head -8
python3 tools/summarize_profile.py gpurun_out/r02 gpurun_out/r02_sum r02 straight | tail -3
python3 tools/summarize_profile.py gpurun_out/r02_loco3d gpurun_out/r02_sum r02_loco3d loco3d | tail -3
python3 tools/summarize_profile.py gpurun_out/r02_policy gpurun_out/r02_sum r02_policy policy | head -12
