#!/bin/bash
# usage (GPU box, repo root): tools/gpu_profiles_r6.sh <tag>  -- round 6's profile stamp: headline + 19-dof walker (gpu_round_profile.sh: kernel trace, PMC passes incl. the dynamic
# VALU mix), the two persistent kernels with the random-init AND the trained (walking) policy (gpu_policy_pmc.sh), tools/diag_walking.py (product and -DDL_EXP_ROLLOUT_PROF=2 build),
# the numpy VecEnv surface; condensed into gpurun_out/<tag>_sum/ (copy to profiles/)
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
S=gpurun_out/${TAG}_sum; mkdir -p $S
tools/gpu_round_profile.sh $TAG straight > /dev/null
tools/gpu_round_profile.sh ${TAG}_loco3d loco3d > /dev/null
python3 tools/summarize_profile.py gpurun_out/$TAG $S $TAG straight | tail -3 | cut -c1-300
python3 tools/summarize_profile.py gpurun_out/${TAG}_loco3d $S ${TAG}_loco3d loco3d | tail -3 | cut -c1-300
tools/gpu_policy_pmc.sh $TAG "policy policy_per_rollout policy_walking policy_walking_per_rollout" > /dev/null
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/diag_walking.py 6 2>&1 | grep -v amdgpu.ids > $S/${TAG}_walking_summary.txt
if [ -f build_variants/libdrloco_hip_prof.so ]; then
  echo "" >> $S/${TAG}_walking_summary.txt; echo "---- the same with the -DDL_EXP_ROLLOUT_PROF=2 build (per-step phase records of the exact mode; the records cost a few per cent):" >> $S/${TAG}_walking_summary.txt
  DL_LIB_PATH=$PWD/build_variants/libdrloco_hip_prof.so timeout 900 python3 tools/diag_walking.py 4 2>&1 | grep -v amdgpu.ids >> $S/${TAG}_walking_summary.txt
fi
timeout 600 python3 tools/bench_vecenv_api.py 2>&1 | grep -v amdgpu.ids > $S/${TAG}_vecenv_api.txt
ls $S
grep -h "VALU busy\|MFMA busy\|FP32 arithmetic\|active lanes" $S/*_summary.txt | cut -c1-220
cp $S/traffic_env_step*.json profiles/ 2>/dev/null
for extra in "" "--walker loco3d" "--policy" "--policy --moments per_rollout" "--policy --checkpoint walking --warmup 8" "--policy --checkpoint walking --warmup 8 --moments per_rollout"; do
  python3 bench.py --no-cpu-baseline $extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$extra', round(d['value']/1e6,2), 'frac', round(r['frac'],5), 'valu_busy', r['valu_busy_frac'], 'mix', (r.get('valu_mix') or {}).get('fp32_frac_of_157_3_tf_vector_peak'), 'traffic', r['traffic'], r['from_profile'])"
done
