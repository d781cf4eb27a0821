#!/bin/bash
# round 6, second GPU pass: the whole -m gpu suite (incl. the hardware probes and quirk Q4), smoke(), the benchmark line on both code objects.
out=gpurun_out/r6b; mkdir -p $out
python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log; tail -15 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
b() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$out/$name.json')); print(round(d['value']/1e6,2),'M', round(d['ms_per_step'],2),'ms', d['code_object'], d['roofline']['from_profile'])" 2>&1 | tail -1)"; }
b auto --steps 20 --warmup 5
DL_DPP_WAIT=2 b w2 --steps 20 --warmup 5
b auto_again --steps 20 --warmup 5
DL_DPP_WAIT=2 b w2_again --steps 20 --warmup 5
b loco3d --walker loco3d --steps 10 --warmup 3
DL_DPP_WAIT=2 b loco3d_w2 --walker loco3d --steps 10 --warmup 3
b policy --policy --steps 10 --warmup 3
DL_DPP_WAIT=2 b policy_w2 --policy --steps 10 --warmup 3
