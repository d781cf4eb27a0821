#!/bin/bash
# usage (GPU box): tools/gpu_split.sh -- the split-workgroup step kernel (DL_SPLIT=1) against the product launch form: tests, placement-free timing
cd $GRAFT_REPO_ROOT
b() { timeout 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))"; }
echo "== DL_SPLIT=1: single-step and rollout tests (float32, 16 lanes)"
DL_SPLIT=1 timeout 600 python3 -m pytest tests -m gpu -q -x -k "single_step_f32 or test_rollout_f32 or multi_step or ragged or G4 or flips" 2>&1 | tail -5
echo -n "product bench: "; b; echo -n "product policy: "; b --policy
echo -n "split bench: "; DL_SPLIT=1 b; echo -n "split policy: "; DL_SPLIT=1 b --policy
echo -n "split 32768: "; DL_SPLIT=1 b --envs-per-gpu 32768
