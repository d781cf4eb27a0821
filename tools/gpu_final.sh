cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 2400 python3 -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; tail -3 gpurun_out/final/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/final/bench.json; python3 -c "
import json; d=json.load(open('gpurun_out/final/bench.json')); r=d['roofline']; print(round(d['value']/1e6,2), r['from_profile'], r['valu_busy_frac'], r['traffic'], r['frac'], d['cpu_baseline']['value'])"
