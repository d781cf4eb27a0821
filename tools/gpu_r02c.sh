#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r02c.sh <tag>  -- policy-in-the-loop path after the LDS-lean policy kernel / fused normalisation, variants
TAG=${1:-r02c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "policy or vecnormalize or rollout or group or overlap or sb3 or gae or config0" > $OUT/pytest_focus.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_focus.log
tail -8 $OUT/pytest_focus.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
run default
run policy --policy
run policy_h2 --policy --handles 2
run policy_8192 --policy --envs-per-gpu 8192
run policy_8192_h2 --policy --envs-per-gpu 8192 --handles 2
run policy_16384_h2 --policy --envs-per-gpu 16384 --handles 2
export DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libdrloco_hip_nopin.so
run nopin_default
run nopin_policy --policy
unset DL_LIB_PATH
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/policy_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy --steps 2 > $OUT/policy_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 > $OUT/default_trace.log 2>&1
python3 - <<PY
import csv, glob
for d in ('policy_trace', 'default_trace'):
    f = glob.glob('$OUT/' + d + '/*/*kernel_stats.csv')
    print(d)
    if f:
        for r in list(csv.DictReader(open(f[0])))[:7]:
            print(f"  {r['Name'][:70]:70s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.2f} us {float(r['Percentage']):6.2f} %")
PY
