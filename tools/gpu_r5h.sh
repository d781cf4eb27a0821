#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/bench.txt
done
python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --walker loco3d 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('loco3d', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/bench.txt
