#!/bin/bash
# usage (GPU box, repo root): tools/gpu_profiles_r5.sh <tag>  -- the round's ONE profile re-stamp: headline + 19-dof walker (gpu_round_profile.sh), both persistent
# kernels (gpu_policy_pmc.sh), config 5; condensed into gpurun_out/<tag>_sum/ (copy to profiles/)
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
tools/gpu_round_profile.sh $TAG straight > /dev/null
tools/gpu_round_profile.sh ${TAG}_loco3d loco3d > /dev/null
S=gpurun_out/${TAG}_sum
python3 tools/summarize_profile.py gpurun_out/$TAG $S $TAG straight | tail -3 | cut -c1-300
python3 tools/summarize_profile.py gpurun_out/${TAG}_loco3d $S ${TAG}_loco3d loco3d | tail -3 | cut -c1-300
tools/gpu_policy_pmc.sh $TAG > /dev/null
R=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_randomize; mkdir -p $R
python3 bench.py --no-cpu-baseline --randomize > $R/bench.json 2> $R/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --randomize > $R/bench_trace.log 2>&1
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --randomize --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/pmc1 -- $B > $R/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/pmc2 -- $B > $R/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_profile.py gpurun_out/${TAG}_randomize $S ${TAG}_randomize randomize > /dev/null
timeout 600 python3 tools/diag_randomize.py 2>&1 | grep -v amdgpu.ids >> $S/${TAG}_randomize_summary.txt
ls $S
grep -h "VALU busy\|MFMA busy\|L2 ->" $S/*_summary.txt | cut -c1-200
# the bench lines with the fresh profiles in place (what the driver's run will print)
cp $S/traffic_env_step*.json profiles/ 2>/dev/null
for extra in "" "--walker loco3d" "--policy" "--policy --moments per_rollout"; do
  python3 bench.py --no-cpu-baseline $extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$extra', round(d['value']/1e6,2), 'frac', round(r['frac'],5), 'valu_busy', r['valu_busy_frac'], 'mfma', r.get('mfma_busy_frac'), 'traffic', r['traffic'], r['from_profile'])"
done
