#!/usr/bin/env python3
"""Split workgroups (dl_set_split; needs a build with -DDL_EXP_SPLIT_PROF, selected with DL_LIB_PATH): cycles the dynamics wave spends in the smooth dynamics and waiting
for its constraint wave, per control step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.vec_env import HipVecEnv
n, T = 4096, 64
if len(sys.argv) > 1 and sys.argv[1] == 'loco3d':          # the 19-dof walker's split workgroups (round 5)
    from drloco_amd import mocap, models
    _ang, _vel = mocap.synthetic_loco3d(L=60000, seed=0)
    env = HipVecEnv(models.WALKER_165CM, num_envs=n, seed=1, refs=mocap.loco3d_table(_ang, _vel), lanes_per_walker="split")
else:
    env = HipVecEnv(num_envs=n, seed=1, lanes_per_walker="split")
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(3)
acts = torch.clamp(0.5 * torch.randn(T, n, env.nu, device='cuda', generator=g), -1, 1)
env.rollout_fixed(acts)              # warm-up: walkers spread over the gait
env.debug_counters()
env.rollout_fixed(acts)
c = env.debug_counters().astype(np.float64) * 16
wait, smooth, total = c[1][::4] / T, c[2][::4] / T, c[3][::4] / T
srv = c[0][::4] / T          # the constraint wave's busy cycles (request seen -> answer posted)
print(f'per control step and dynamics wave (cycles): whole {total.mean():.0f} (max {total.max():.0f}), smooth dynamics {smooth.mean():.0f}, waiting for the constraint wave {wait.mean():.0f} (max {wait.max():.0f}); constraint wave busy {srv.mean():.0f} per step = {srv.mean() / 20:.0f} per evaluation')
