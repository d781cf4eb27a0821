#!/usr/bin/env python3
"""How many contacts does an evaluation of the 19-dof walker (BASELINE config 4) carry?  The LDS question behind a look-ahead split for that walker
(VERDICT r4 item 3): the contact Jacobians take 24 x 16 x 4 words of the 9.98 KB per walker; a split workgroup of sixteen walkers has 10 KB per walker
in all.  The benchmark's rollout (4096 walkers, bench.py's action noise, synthetic loco3d table, mixed-clip RSI), sampled with dl_forward at the state
after every 4th control step: histogram of ncon (active contacts) and of the constraint rows (limits + 4 per contact).
usage: python3 tools/diag_ncon_hist.py [straight]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from drloco_amd import mocap, models
from drloco_amd.vec_env import HipVecEnv

n, T = 4096, 512
straight = len(sys.argv) > 1 and sys.argv[1] == 'straight'
if straight:
    env = HipVecEnv(num_envs=n, seed=1234)
else:
    ang, vel = mocap.synthetic_loco3d(L=60000, seed=0)
    env = HipVecEnv(models.WALKER_165CM, num_envs=n, seed=1234, refs=mocap.loco3d_table(ang, vel))
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
maxc = 18 if straight else 24
hist = np.zeros(maxc + 1, np.int64); rows_hist = np.zeros(130, np.int64)
samples = 0
for t in range(T):
    a = torch.clamp(0.5 * torch.randn(n, env.nu, device='cuda', generator=g), -1, 1)
    env.step_tensors(a)
    if t % 4 == 3:
        _, ncon, nefc, _ = env.forward()
        hist += np.bincount(np.clip(ncon, 0, maxc), minlength=maxc + 1)
        rows_hist += np.bincount(np.clip(nefc, 0, 129), minlength=130)
        samples += n
cum = np.cumsum(hist) / samples
print(('straight walker' if straight else '19-dof walker') + f': {samples} sampled evaluations ({n} walkers, every 4th of {T} control steps)')
print('ncon   count      fraction   cumulative')
for c in range(maxc + 1):
    if hist[c]:
        print(f'{c:4d} {hist[c]:9d}   {hist[c] / samples:9.5f}   {cum[c]:9.5f}')
for lim in (8, 12, 16, 20):
    if lim <= maxc:
        print(f'evaluations with more than {lim} contacts: {1 - cum[lim]:.5f}')
r = np.cumsum(rows_hist) / samples
print('constraint rows (limits + 4 per contact): mean %.1f, p50 %d, p99 %d, max %d' % ((rows_hist * np.arange(130)).sum() / samples, int(np.searchsorted(r, 0.5)), int(np.searchsorted(r, 0.99)), int(np.nonzero(rows_hist)[0].max())))
