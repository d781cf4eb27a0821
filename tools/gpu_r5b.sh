#!/bin/bash
# round 5: tied 16x16x4 MFMAs -- policy / persistent tests incl. the soak, the chain ubench, the three --policy bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_persistent.py tests/test_gpu_bench_shapes.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "policy" > $O/pytest_policy.log 2>&1; echo "pytest rc=$?" >> $O/pytest_policy.log
tail -3 $O/pytest_policy.log
hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_overlap_chain.hip -o /tmp/moc 2>/dev/null && timeout 300 /tmp/moc > $O/mfma_overlap_chain.txt 2>&1; cat $O/mfma_overlap_chain.txt
for extra in "" "--policy" "--policy --moments per_rollout" "--policy --rollout-form launches"; do
  python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 $extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$extra', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/bench.txt
done
