#!/usr/bin/env python3
"""Solver diagnostics on the GPU: distribution of Newton iterations / constraint rows / diverged steps per
control step of the 16-lane kernel over a rollout, and per-launch durations."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drloco_amd.vec_env import HipVecEnv

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--precision', type=int, default=32)
args = ap.parse_args()
env = HipVecEnv(num_envs=args.envs, lanes_per_walker=16, precision=args.precision, seed=1234)
env.reset_tensors()
env.debug_counters()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(args.steps, args.envs, 8, device='cuda', generator=g), -1, 1)
tot_div = 0
ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
rows = []
for t in range(args.steps):
    ev[t].record()
    _, _, done, _ = env.step_tensors(acts[t])
    c = env.debug_counters()
    it, mx, nr, dv = c
    # per wave (4 consecutive walkers): the wave iterates as long as its slowest walker
    rows.append((t, it.mean(), it.max(), mx.max(), nr.mean() / 20, int(dv.sum()), int(done.sum().item())))
    tot_div += int(dv.sum())
    if dv.sum():
        w = np.nonzero(dv)[0]
        st = env.get_state()
        print(f'step {t}: diverged walkers {w[:8]} iters {it[w][:8]} cursor ep_dur {st["cursor"][4, w][:8]}')
ev[args.steps].record()
torch.cuda.synchronize()
ms = np.array([ev[t].elapsed_time(ev[t + 1]) for t in range(args.steps)])
print('t  mean_iters/step  max_iters/step  max_iters/eval  rows/eval  diverged  done   ms')
for r, m in zip(rows, ms):
    if r[0] % 10 == 0 or r[2] > 150 or r[5]:
        print('%3d  %8.1f  %6d  %5d  %6.1f  %3d  %4d  %7.3f' % (*r, m))
print('total diverged', tot_div, ' step ms: median %.3f mean %.3f max %.3f' % (np.median(ms), ms.mean(), ms.max()))
