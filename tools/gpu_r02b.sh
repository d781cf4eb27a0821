#!/bin/bash
# usage (GPU box, repo root): tools/gpu_r02b.sh <tag>  -- full GPU tests, bench variants, kernel trace of the policy-in-the-loop run
TAG=${1:-r02b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
for v in "" "--policy" "--randomize" "--walker loco3d" "--walker loco3d --policy" "--envs-per-gpu 8192" "--envs-per-gpu 8192 --policy"; do
  name=$(echo "bench$v" | tr ' ' '_' | tr -d '-')
  timeout 600 python3 bench.py --no-cpu-baseline $v > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']/1e6,2), 'M env-steps/s', round(d['roofline']['avg_launch_us'],1), 'us/launch')" || tail -3 $OUT/$name.err
done
python3 tools/bench_gae.py > $OUT/bench_gae.txt 2>&1; cat $OUT/bench_gae.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/policy_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --policy --steps 2 > $OUT/policy_trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/policy_trace/*/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:10]:
        print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.2f} us {float(r['Percentage']):6.2f} %")
PY
