#!/usr/bin/env python3
"""Which evaluations make the tail of the step kernel?  Captures stage inputs of evaluations that needed >= K Newton
iterations in the float32 16-lane kernel (DL_DEBUG_CAP_ITERS=K) and replays them through dl_forward in float32 and
float64: if float64 needs as many iterations the active-set search is genuinely long, if not the float32 stopping tests
chase rounding noise."""
import argparse, os, sys
ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=5)
ap.add_argument('--steps', type=int, default=60)
args = ap.parse_args()
os.environ['DL_DEBUG_CAP_ITERS'] = str(args.k)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drloco_amd.vec_env import HipVecEnv
from drloco_amd import abi

n = 4096
env = HipVecEnv(num_envs=n, lanes_per_walker=16, seed=1234)
env.reset_tensors(); env.debug_counters()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(80 + args.steps, n, 8, device='cuda', generator=g), -1, 1)
for t in range(80):
    env.step_tensors(acts[t])
env.debug_counters()
cases = []
for t in range(80, 80 + args.steps):
    pre = env.get_state()
    env.step_tensors(acts[t])
    it, mx, nr, dv = env.debug_counters()
    hit = np.nonzero(mx >= args.k)[0]
    if len(hit):
        cs = env.debug_capstate()
        left = np.asarray(env.refs.step_is_left)[pre['cursor'][abi.DL_CUR_I_STEP, hit]]
        for w, l in zip(hit, left):
            u = 300.0 * np.clip(acts[t, w].cpu().numpy().astype(np.float64), -1, 1)
            if l:
                u = u[[4, 5, 6, 7, 0, 1, 2, 3]]; u[1] = -u[1]; u[5] = -u[5]
            cases.append((cs[0:14, w].copy(), cs[16:30, w].copy(), cs[32:46, w].copy(), u, int(mx[w])))
print(f'{len(cases)} walker-steps with an evaluation of >= {args.k} iterations in {args.steps} steps x {n} walkers '
      f'({100.0 * len(cases) / (args.steps * n):.2f} % of walker-steps)')
cases = cases[:2048]
m = len(cases)
if m:
    Q = np.stack([c[0] for c in cases], 1).astype(np.float64); V = np.stack([c[1] for c in cases], 1).astype(np.float64)
    W = np.stack([c[2] for c in cases], 1).astype(np.float64); U = np.stack([c[3] for c in cases], 1)
    res = {}
    for name, prec in (('f32', 32), ('f64', 64)):
        e = HipVecEnv(num_envs=m, lanes_per_walker=16, precision=prec)
        e.set_state(qpos=Q, qvel=V, warm=W)
        qa, nc, ne, ni = e.forward(U)
        res[name] = (qa, nc, ne, ni)
        print(f'{name}: iterations hist {np.bincount(ni)[:16]}  mean {ni.mean():.2f}   ncon mean {nc.mean():.1f}  nefc mean {ne.mean():.1f}')
    d = np.abs(res['f32'][0] - res['f64'][0]).max(0) / (np.abs(res['f64'][0]).max(0) + 1e-9)
    print('relative acceleration difference f32 vs f64: median %.2e  p90 %.2e  max %.2e' % (np.median(d), np.percentile(d, 90), d.max()))
    print('recorded max iterations of those steps: hist', np.bincount([c[4] for c in cases])[:16])
    both = np.stack([res['f32'][3], res['f64'][3]], 1)
    print('f32 iterations -> mean f64 iterations:', {int(k): round(float(both[both[:, 0] == k, 1].mean()), 2) for k in np.unique(both[:, 0])})
