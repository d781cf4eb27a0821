#!/usr/bin/env python3
"""Soak run: R x 500 control steps of 4096 walkers under random torques through dl_rollout_fixed; checks that every output stays finite
and in range and prints the solver statistics and the number of walker-steps that took the exception path (physics divergence ->
episode ends with reward 0, mimic_env.py:86-91).  SOAK_PREC=64 runs the float64 build of the same kernels for comparison."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from drloco_amd.vec_env import HipVecEnv
import os
n, T, R = 4096, 500, int(os.environ.get("SOAK_R", "10"))
PREC = int(os.environ.get("SOAK_PREC", "32"))
_kw = {'lanes_per_walker': 'split'} if os.environ.get('SOAK_SPLIT') == '1' else {}      # SOAK_SPLIT=1: the split-workgroup launch form
if os.environ.get('SOAK_WALKER') == 'loco3d':          # the 19-dof walker (round 5: also in the split form)
    from drloco_amd import mocap, models
    _ang, _vel = mocap.synthetic_loco3d(L=60000, seed=0)
    env = HipVecEnv(models.WALKER_165CM, num_envs=n, seed=4242, precision=PREC, refs=mocap.loco3d_table(_ang, _vel), **_kw)
else:
    env = HipVecEnv(num_envs=n, seed=4242, precision=PREC, **_kw)
env.reset_tensors(); env.debug_counters()
g = torch.Generator(device='cuda'); g.manual_seed(1)
tot_done = 0
for r in range(R):
    acts = torch.clamp(0.5 * torch.randn(T, n, env.nu, device='cuda', generator=g), -1, 1)
    obs, rew, done = env.rollout_fixed(acts)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all(), r
    assert float(rew.max()) <= 1.2 + 1e-5 and float(rew.min()) >= 0.0, (float(rew.min()), float(rew.max()))
    tot_done += int(done.sum())
    it, mx, rows, div = env.debug_counters()
    print(f'rollout {r}: episodes ended {int(done.sum())}, mean reward {float(rew.mean()):.3f}, iterations/step mean {it.mean() / T:.1f}, max per evaluation {mx.max()}, diverged walker-steps {int(div.sum())}', flush=True)
st = env.get_state()
assert np.isfinite(st['qpos']).all() and np.isfinite(st['qvel']).all()
print('ok', tot_done)
