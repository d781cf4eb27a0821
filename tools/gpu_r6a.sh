#!/bin/bash
# round 6, first GPU pass: quirk Q4 on the device (golden G15), the walking workload (bench.py --policy --checkpoint walking) next to the random-init policy,
# tools/diag_walking.py, the numpy VecEnv surface.  Outputs under gpurun_out/r6a/.
out=gpurun_out/r6a; mkdir -p $out
python -m pytest tests -m gpu -q -x -k "G15 or ref_offsets or one_call_evaluation or G4_step or G3_reward" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -3 $out/tests.log
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$out/$name.json')); print(round(d['value']/1e6,2),'M', round(d['ms_per_step'],2),'ms', d['self_check'].get('solver'), d['self_check'].get('walking'))" 2>&1 | tail -1)"; }
b headline --steps 10 --warmup 3
b walk_exact --policy --checkpoint walking --warmup 8 --steps 10
b walk_exact_stats --policy --checkpoint walking --warmup 8 --steps 10 --solver-stats
b walk_perrollout --policy --checkpoint walking --warmup 8 --steps 10 --moments per_rollout
b walk_launches --policy --checkpoint walking --warmup 8 --steps 6 --rollout-form launches
b walk_det --policy --checkpoint walking --warmup 8 --steps 10 --deterministic
b rand_exact --policy --warmup 8 --steps 10
b rand_exact_stats --policy --warmup 8 --steps 10 --solver-stats
b rand_perrollout --policy --warmup 8 --steps 10 --moments per_rollout
timeout 900 python tools/diag_walking.py 4 > $out/diag_walking.txt 2>&1; tail -30 $out/diag_walking.txt
timeout 600 python tools/bench_vecenv_api.py > $out/vecenv_api.txt 2>&1; cat $out/vecenv_api.txt
