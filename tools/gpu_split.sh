#!/bin/bash
# usage (GPU box): tools/gpu_split.sh -- the split-workgroup step kernel (dl_set_split) against the one-wave launch form: tests and benchmark lines
cd $GRAFT_REPO_ROOT
b() { timeout 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))"; }
echo "== split workgroups: the float32 tests of the straight walker that run both launch forms"
timeout 900 python3 -m pytest tests -m gpu -q -x -k "split" 2>&1 | tail -3
echo -n "one wave per four walkers: "; b --no-split; echo -n "  with a policy: "; b --policy --no-split
echo -n "split workgroups:          "; b; echo -n "  with a policy: "; b --policy
echo -n "split workgroups, 32768 walkers: "; b --envs-per-gpu 32768
