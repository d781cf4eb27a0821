#!/bin/bash
# usage (GPU box, repo root): tools/gpu_policy_trace.sh  -- GPU-side durations of dl_policy_forward by batch size
D=/tmp/pp_$$_$RANDOM
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/bench_policy.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$D/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_policy_forward" in r["Kernel_Name"]:
        d[(int(r["Grid_Size_X"]) // 256, int(r["Dispatch_Id"]) // 220)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        meta = r
for k in sorted(d):
    v = sorted(d[k]); us = v[len(v) // 2]
    print("%6d rows (batch of launches %d): median %.1f us  min %.1f" % (k[0] * 16, k[1], us, v[0]))
print("vgpr", meta["VGPR_Count"], "agpr", meta["Accum_VGPR_Count"], "scratch", meta["Scratch_Size"])
PY
