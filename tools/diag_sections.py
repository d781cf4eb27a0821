#!/usr/bin/env python3
"""Per-section cycle counts of one forward evaluation (16-lane f32 kernel) at mid-rollout states."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drloco_amd import lib
from drloco_amd.vec_env import HipVecEnv, _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--warm', type=int, default=80)
ap.add_argument('--walker', choices=['straight', 'loco3d'], default='straight')
args = ap.parse_args()
n = args.envs
if args.walker == 'loco3d':
    from drloco_amd import mocap, models
    ang, vel = mocap.synthetic_loco3d(L=60000, seed=0)
    env = HipVecEnv(models.WALKER_165CM, num_envs=n, lanes_per_walker=16, seed=1234, refs=mocap.loco3d_table(ang, vel))
else:
    env = HipVecEnv(num_envs=n, lanes_per_walker=16, seed=1234)
nu, nv = env.nu, env.nv
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(args.warm, n, nu, device='cuda', generator=g), -1, 1)
for t in range(args.warm):
    env.step_tensors(acts[t])
ctrl = (300 * torch.clamp(0.5 * torch.randn(nu, n, device='cuda', generator=g), -1, 1)).contiguous()
nb = (n + 3) // 4
tim = torch.zeros(10, nb, dtype=torch.int64, device='cuda')
qacc = torch.zeros(nv, n, device='cuda')
names = ['smooth dynamics', 'constraints', 'rows, J^T f, Hessian', 'factor + solve', 'J dir / M dir', 'line search + step', '(iterations)', '-']
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.check(env._lib.dl_debug_forward_timed(env._h, _ptr(ctrl), _ptr(qacc), _ptr(tim), _stream()))
    e1.record(); torch.cuda.synchronize()
    t = tim.cpu().numpy().astype(np.float64)[:8]
tot = t[:6].sum(0)
print(f'kernel {e0.elapsed_time(e1) * 1e3:.1f} us; per wave: total cycles mean {tot.mean():.0f} median {np.median(tot):.0f} max {tot.max():.0f}; wave iterations mean {t[6].mean():.2f} max {t[6].max():.0f}')
for k in range(6):
    print(f'  {names[k]:22s} mean {t[k].mean():9.0f}  ({100 * t[k].mean() / tot.mean():5.1f} %)  max {t[k].max():9.0f}   per iteration {t[k].mean() / t[6].mean():8.0f}')
if t[7].sum() > 0:      # a -DDL_EXP_LS_COUNT build: trials * 65536 + line searches per wave
    ls, tr = np.mod(t[7], 65536), np.floor(t[7] / 65536)
    print(f'line searches per wave and evaluation: mean {ls.mean():.2f}; trials per line search: {tr.sum() / max(1, ls.sum()):.2f}; trials per wave mean {tr.mean():.2f} max {tr.max():.0f}')
qa, nc, ne, ni = env.forward(ctrl.cpu().numpy().astype(np.float64))
print('iters: mean %.2f  hist %s' % (ni.mean(), np.bincount(ni)[:12]))
print('nefc: mean %.1f  hist(0,1-8,9-16,17-32,33+) %s' % (ne.mean(), [int((ne == 0).sum()), int(((ne > 0) & (ne <= 8)).sum()), int(((ne > 8) & (ne <= 16)).sum()), int(((ne > 16) & (ne <= 32)).sum()), int((ne > 32).sum())]))
w = ni.reshape(-1, 4).max(1)
print('per wave max iters: mean %.2f' % w.mean())

# ---- the same sections inside whole control steps of the rollout (20 evaluations each)
acts2 = torch.clamp(0.5 * torch.randn(30, n, nu, device='cuda', generator=g), -1, 1)
tot = np.zeros((10, nb))
launch_max, launch_mean, per_launch = [], [], []
for t in range(30):
    lib.check(env._lib.dl_debug_step_timed(env._h, _ptr(acts2[t]), _ptr(env.obs), _ptr(env.rew), _ptr(env.done), _ptr(tim), _stream()))
    torch.cuda.synchronize()
    tot += tim.cpu().numpy().astype(np.float64)
    launch_max.append(float(tim[7].max())); launch_mean.append(float(tim[7].double().mean())); per_launch.append(tim[7].cpu().numpy().astype(np.float64))
tot /= 30
whole = tot[7]
print(f'control step: per wave cycles mean {whole.mean():.0f} median {np.median(whole):.0f} max {whole.max():.0f} (the launch lasts as long as its slowest wave); wave iterations per step mean {tot[6].mean():.1f} max {tot[6].max():.0f}')
print('  wave-time percentiles (cycles): ' + '  '.join(f'p{q} {np.percentile(whole, q):.0f}' for q in (10, 50, 90, 99, 100)))
for k in range(6):
    print(f'  {names[k]:22s} mean {tot[k].mean():9.0f}  ({100 * tot[k].mean() / whole.mean():5.1f} %)  slowest wave {tot[k][whole.argmax()]:9.0f}')
print(f'  {"outside the evaluations":22s} mean {(whole - tot[:6].sum(0)).mean():9.0f}  ({100 * (whole - tot[:6].sum(0)).mean() / whole.mean():5.1f} %)')
print(f'    of which: before the physics (model/state/action loads) {tot[8].mean():9.0f}   after it (cursor, reward, observation, Monitor, reset, stores) {tot[9].mean():9.0f}   inside the RK4 loop {(whole - tot[:6].sum(0) - tot[8] - tot[9]).mean():9.0f}')
print(f'per launch: slowest wave mean over launches {np.mean(launch_max):.0f} (min {np.min(launch_max):.0f}, max {np.max(launch_max):.0f}); mean wave {np.mean(launch_mean):.0f}; '
      f'slowest / mean = {np.mean(launch_max) / np.mean(launch_mean):.3f}; slowest of the 30-step averages / mean = {whole.max() / whole.mean():.3f}')
P = np.stack(per_launch)                                  # [launches, waves]
for K in (2, 4, 8, 16):
    # a launch of K control steps would last as long as the wave with the largest K-step sum
    m = [P[i:i + K].sum(0).max() / K for i in range(0, P.shape[0] - K + 1, K)]
    print(f'  {K:2d} control steps per launch: slowest wave per control step {np.mean(m):.0f}  ({np.mean(m) / np.mean(launch_mean):.3f} x the mean wave)')
