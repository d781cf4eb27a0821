#!/bin/bash
# usage: tools/gpu_quick.sh  -- focused GPU tests + the benchmark line three times + policy
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "forward or single_step or rollout_f64 or full_size or randomization" > gpurun_out/quick/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/quick/pytest.log
tail -3 gpurun_out/quick/pytest.log
for i in 1 2 3; do python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))"; done
python3 bench.py --no-cpu-baseline --policy 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('policy', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))"
python3 bench.py --no-cpu-baseline --walker loco3d 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('loco3d', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))"
