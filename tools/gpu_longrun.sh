#!/bin/bash
# usage (GPU box): tools/gpu_longrun.sh -- long timed regions of the main bench forms (hundreds of rollouts: hand-over / exchange time-outs, faults and the self-check would show)
cd $GRAFT_REPO_ROOT
out=gpurun_out/longrun; mkdir -p $out
run() { name=$1; shift; timeout 1200 python3 bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python3 -c "import json; d=json.load(open('$out/$name.json')); sc=d['self_check']; print('$name', round(d['value']/1e6,2), 'M', d['steps'], 'rollouts', {k: sc.get(k) for k in ('finite','exception_path_steps','walker_steps_run','episodes_ended_last_rollout')})" || tail -3 $out/$name.err; }
run default --steps 400 --warmup 2
run policy_exact --policy --steps 150 --warmup 2
run policy_per_rollout --policy --moments per_rollout --steps 200 --warmup 2
run walking --policy --checkpoint walking --steps 150 --warmup 8
run walking_per_rollout --policy --checkpoint walking --moments per_rollout --steps 200 --warmup 8
run loco3d --walker loco3d --steps 150 --warmup 2
run loco3d_policy --walker loco3d --policy --steps 60 --warmup 2
run loco3d_policy_per_rollout --walker loco3d --policy --moments per_rollout --steps 80 --warmup 2
