#!/usr/bin/env python3
"""BASELINE config 5 ("divergent-contact stress") against the nominal dynamics at the benchmark's size and launch form: what the per-walker mass / friction
randomisation and the 50 N push schedule do to the solver's work.  4096 walkers, split workgroups, bench.py's action noise and its randomisation (keyed by
the walker index), 256 control steps in launches of 64: Newton iterations and constraint rows per forward evaluation (dl_debug_counters), episode ends,
and the time of the step launches (HIP events).  usage: python3 tools/diag_randomize.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from drloco_amd.vec_env import HipVecEnv

n, T, chunk = 4096, 256, 64
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(T, n, 8, device='cuda', generator=g), -1, 1)
rows = []
for label in ('(first handle of the process: warm-up, discarded)', 'nominal', 'randomised + pushes'):
    env = HipVecEnv(num_envs=n, seed=1234, lanes_per_walker='split')
    if label.startswith('randomised'):
        gidx = np.arange(n)
        u = lambda salt: np.array([np.random.default_rng((int(i), salt)).random() for i in gidx])
        env.set_randomization(0.8 + 0.4 * u(1), 0.5 + 0.6 * u(2))
        ang = 2 * np.pi * u(3)
        env.set_push_schedule(np.stack([50 * np.cos(ang), 50 * np.sin(ang), 0 * ang], 1), (400 * u(4)).astype(np.int32), period=400, duration=20)
    env.reset_tensors()
    env.debug_counters()
    env.rollout_fixed(acts[:chunk])                      # warm-up launch (every walker starts a fresh episode)
    env.debug_counters(clear=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dones = 0
    ev0.record()
    for t0 in range(0, T, chunk):
        _, _, d = env.rollout_fixed(acts[t0:t0 + chunk])
        dones += int(d.sum())
    ev1.record(); torch.cuda.synchronize()
    c = env.debug_counters(clear=False).astype(np.float64)          # [4, n]: sum iterations, max of the last step, sum rows, diverged steps
    evals = T * 4 * env.model.frame_skip
    rows.append((label, ev0.elapsed_time(ev1) / T * 1e3, c[0].mean() / evals, np.quantile(c[0] / evals, 0.99), c[2].mean() / evals, int(c[3].sum()), dones))
    env.close()
print(f'{n} walkers x {T} control steps, split workgroups, launches of {chunk} steps')
print(f'{"dynamics":22s} {"us / control step":>18s} {"iterations / eval":>18s} {"(p99 walker)":>13s} {"rows / eval":>12s} {"diverged steps":>15s} {"episode ends":>13s}')
for r in rows[1:]:
    print(f'{r[0]:22s} {r[1]:18.1f} {r[2]:18.3f} {r[3]:13.3f} {r[4]:12.2f} {r[5]:15d} {r[6]:13d}')
rows = rows[1:]
a, b = rows
print(f'randomised / nominal: time {b[1] / a[1]:.3f}, iterations {b[2] / a[2]:.3f}, rows {b[4] / a[4]:.3f}')
