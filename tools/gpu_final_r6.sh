#!/bin/bash
# round 6's closing GPU pass: the whole -m gpu suite, smoke(), the profile stamp (tools/gpu_profiles_r6.sh) and every line of the tables (tools/gpu_table.sh)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6final; mkdir -p $out
python -m pytest tests -m gpu -q > $out/tests.log 2>&1; echo "gpu suite rc=$?"; tail -4 $out/tests.log
for i in 2 3; do python -m pytest tests -m gpu -q -p no:cacheprovider > $out/tests_pass$i.log 2>&1; echo "gpu suite, pass $i: rc=$?"; tail -1 $out/tests_pass$i.log; done          # (flaky-failure watch: the same suite twice more)
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log | cut -c1-200
bash tools/gpu_profiles_r6.sh r06 > $out/profiles.log 2>&1; tail -12 $out/profiles.log | cut -c1-260
bash tools/gpu_table.sh r06table > $out/table.txt 2>&1; cat $out/table.txt
python bench.py --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err; python -c "import json; d=json.load(open('$out/bench_driver_form.json')); print('driver form:', round(d['value']/1e6,2), d['roofline']['frac'], d['roofline']['valu_busy_frac'], d['roofline']['from_profile'], d['code_object'], d['self_check'])" | cut -c1-900
