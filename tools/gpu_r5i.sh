#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_persistent.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log | cut -c1-300
for extra in "--walker loco3d --policy" "--walker loco3d --policy --moments per_rollout" "--walker loco3d --policy --rollout-form launches" "--walker loco3d"; do
  timeout 600 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 $extra 2>$O/bench.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$extra', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1), d['roofline']['kernel'])" | tee -a $O/bench.txt || tail -5 $O/bench.err
done
