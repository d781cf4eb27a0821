#!/bin/bash
TAG=${1:-r02d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "vecnormalize or overlap or steps_fixed or rollout_fixed or full_size or config0" > $OUT/pytest_focus.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_focus.log
tail -8 $OUT/pytest_focus.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err; }
run default
run randomize --randomize
run loco3d --walker loco3d
run default_8192 --envs-per-gpu 8192
run default_16384 --envs-per-gpu 16384
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 > $OUT/default_trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/default_trace/*/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:7]:
        print(f"  {r['Name'][:70]:70s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.2f} us {float(r['Percentage']):6.2f} %")
PY
