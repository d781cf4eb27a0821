#!/bin/bash
# usage (GPU box, repo root): tools/gpu_policy_profile.sh <tag>  -- bench.py --policy: kernel stats + MFMA counters of k_policy_forward
TAG=${1:-policy}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --policy --no-cpu-baseline --steps 2 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_policy.py > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
out = []
st = glob.glob("$OUT/trace/*/*kernel_stats.csv")
if st:
    out.append("rocprofv3 --kernel-trace --stats -- python3 bench.py --policy --no-cpu-baseline --steps 2   (kernel_stats.csv, top rows)")
    for r in list(csv.DictReader(open(st[0])))[:6]:
        out.append("%-72s calls %6s  avg %10.2f us  %6.2f %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
pm = glob.glob("$OUT/pmc/*/*counter_collection.csv")
if pm:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pm[0])):
        if "k_policy_forward" in r["Kernel_Name"]:
            acc[int(r["Grid_Size"]) // 512 * 16 if "Grid_Size" in r else 0][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append("")
    out.append("PMC of k_policy_forward (tools/bench_policy.py; average per launch, by batch rows):")
    for rows in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[rows].items()}
        out.append("  rows %6d: " % rows + "  ".join("%s %.3g" % (k, c[k]) for k in sorted(c)))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
            out.append("               MFMA busy / SQ busy cycles = %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]))
print("\n".join(out))
open("$OUT/summary.txt", "w").write("\n".join(out) + "\n")
PY
