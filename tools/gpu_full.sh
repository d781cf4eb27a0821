#!/bin/bash
# usage: tools/gpu_full.sh <tag>      the whole -m gpu suite + smoke + the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-full}; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -14 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench.json
python3 -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1), d['roofline']['from_profile'])"
