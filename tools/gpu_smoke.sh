#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
python3 bench.py 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', round(d['value']/1e6,2), d['roofline']['from_profile'], d['roofline']['valu_busy_frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
