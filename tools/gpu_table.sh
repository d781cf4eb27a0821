#!/bin/bash
# usage (GPU box): tools/gpu_table.sh -- the benchmark lines of DESIGN.md 5's table with the product build
cd $GRAFT_REPO_ROOT
b() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],2), 'ms/rollout', round(d['roofline']['avg_launch_us'],1), 'us/launch')"; }
for a in "" "--no-split" "--no-split --vn-single-steps" "--envs-per-gpu 8192" "--envs-per-gpu 32768" "--randomize" "--walker loco3d" "--walker loco3d --envs-per-gpu 16384" "--policy" "--policy --no-split" "--policy --envs-per-gpu 8192" "--policy --envs-per-gpu 8192 --handles 2" "--policy --envs-per-gpu 32768" "--policy --envs-per-gpu 32768 --handles 2 --steps 2" "--policy --envs-per-gpu 65536 --handles 4 --steps 2" "--walker loco3d --policy" "--walker loco3d --policy --envs-per-gpu 16384"; do echo -n "bench.py $a: "; b $a; done
