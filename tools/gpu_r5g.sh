#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q -s -k "loco3d" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "19-dof|loco3d f32|passed|failed|rc=|Error|assert|Fault" $O/pytest.log | cut -c1-300 | tail -20
for extra in "--walker loco3d" "--walker loco3d --no-split" ""; do
  timeout 600 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 $extra 2>$O/bench.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$extra', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1), d['roofline']['kernel'])" | tee -a $O/bench.txt || tail -5 $O/bench.err
done
