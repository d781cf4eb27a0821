#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent.py -m gpu -q -k "f32_randomization or f32_error_growth or G14 or grid_exchange" -s > gpurun_out/r4b/tests.log 2>&1
grep -n "config 5\|float32 vs\|^   [0-9]\|^E  \|passed\|failed" gpurun_out/r4b/tests.log | head -60
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
