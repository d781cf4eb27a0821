#!/usr/bin/env python3
"""A/B bit comparison of two builds of the library: runs the same float32 rollouts (both walkers) with the library named by DL_LIB_PATH
(or the product build) and writes observations / rewards / dones to an .npz; `--compare a.npz b.npz` reports whether they are bit-identical.
usage: DL_LIB_PATH=... tools/ab_bits.py out.npz ;  tools/ab_bits.py --compare a.npz b.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == '--compare':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()
        first = ''
        if a[k].ndim == 3 or (a[k].ndim == 2 and k.endswith(('_rew', '_done'))):          # [T, ...]: the first control step alone (before chaos amplifies the last bit)
            first = f'   first step: max abs diff {np.abs(a[k][0].astype(np.float64) - b[k][0].astype(np.float64)).max():.3e}'
        print(f'{k:24s} identical={same}  max abs diff {d:.3e}{first}')
        ok &= same
    sys.exit(0 if ok else 1)
import torch
from drloco_amd import mocap, models
from drloco_amd.vec_env import HipVecEnv
out = {}
for name in ('straight', 'loco3d'):
    n, T = 512, 60
    if name == 'loco3d':
        ang, vel = mocap.synthetic_loco3d(L=6000, seed=0)
        env = HipVecEnv(models.WALKER_165CM, num_envs=n, lanes_per_walker=16, seed=7, refs=mocap.loco3d_table(ang, vel))
    else:
        env = HipVecEnv(num_envs=n, lanes_per_walker=16, seed=7)
    env.reset_tensors()
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    acts = torch.clamp(0.5 * torch.randn(T, n, env.nu, device='cuda', generator=g), -1, 1)
    O, R, D = [], [], []
    for t in range(T):
        env.step_tensors(acts[t])
        O.append(env.obs.cpu().numpy().copy()); R.append(env.rew.cpu().numpy().copy()); D.append(env.done.cpu().numpy().copy())
    out[name + '_obs'] = np.stack(O); out[name + '_rew'] = np.stack(R); out[name + '_done'] = np.stack(D)
    st = env.get_state()
    out[name + '_qpos'] = np.asarray(st['qpos']); out[name + '_qvel'] = np.asarray(st['qvel'])
np.savez(sys.argv[1], **out)
print('wrote', sys.argv[1])
