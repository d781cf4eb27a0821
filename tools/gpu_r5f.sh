#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
DL_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libdrloco_hip_prof.so timeout 600 python3 tools/diag_rollout_floor.py > $O/rollout_floor.txt 2>&1; grep -v amdgpu.ids $O/rollout_floor.txt | cut -c1-330
timeout 900 python3 tools/diag_ncon_hist.py > $O/ncon_hist_loco3d.txt 2>&1; grep -v amdgpu.ids $O/ncon_hist_loco3d.txt
timeout 900 python3 tools/diag_ncon_hist.py straight > $O/ncon_hist_straight.txt 2>&1; grep -v amdgpu.ids $O/ncon_hist_straight.txt | tail -8
for extra in "--randomize" "--walker loco3d"; do
  python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 $extra 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$extra', round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1))" | tee -a $O/bench.txt
done
