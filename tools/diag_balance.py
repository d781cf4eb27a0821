#!/usr/bin/env python3
"""How much of the step kernel's tail is grouping?  Per-walker solver iterations per control step (debug counters of the
16-lane kernel) at mid-rollout: the iterations of a wave are at least the maximum over its four walkers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drloco_amd.vec_env import HipVecEnv

n = 4096
env = HipVecEnv(num_envs=n, lanes_per_walker=16, seed=1234)
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(120, n, 8, device='cuda', generator=g), -1, 1)
for t in range(80):
    env.step_tensors(acts[t])
env.debug_counters(clear=True)
per = []
for t in range(80, 120):
    env.step_tensors(acts[t])
    c = env.debug_counters(clear=True)
    per.append((c[0].copy(), c[2].copy()))
it = np.stack([p[0] for p in per]).astype(np.float64)      # [steps, walkers] iterations per control step
rows = np.stack([p[1] for p in per]).astype(np.float64)    # summed constraint rows over the 20 evaluations
print('per walker iterations/step: mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % (it.mean(), *np.percentile(it, [50, 90, 99]), it.max()))
print('per walker rows/evaluation:  mean %.1f  p90 %.1f  p99 %.1f  max %.1f' % ((rows / 20).mean(), *np.percentile(rows / 20, [90, 99]), (rows / 20).max()))
w = it.reshape(it.shape[0], -1, 4)
print('max over the 4 walkers of a wave (lower bound of the wave\'s iterations): mean %.1f  max %.0f' % (w.max(2).mean(), w.max(2).max()))
order = np.argsort(it, axis=1)
ws = np.take_along_axis(it, order, 1).reshape(it.shape[0], -1, 4)
print('same after sorting walkers by their own count:                          mean %.1f  max %.0f' % (ws.max(2).mean(), ws.max(2).max()))
# persistence: does a walker's count at step t predict t + 1?
c = np.corrcoef(it[:-1].ravel(), it[1:].ravel())[0, 1]
print('correlation of a walker\'s iterations at step t and t + 1: %.2f ; rows: %.2f' % (c, np.corrcoef(rows[:-1].ravel(), rows[1:].ravel())[0, 1]))
