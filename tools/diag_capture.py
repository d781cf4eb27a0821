#!/usr/bin/env python3
"""Find evaluations where the f32 16-lane solver hits the iteration cap (the step kernel records the stage
input), replay exactly that state through dl_forward with every kernel family and dump it."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drloco_amd.vec_env import HipVecEnv
from drloco_amd import abi

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--out', default='gpurun_out/capture.npz')
args = ap.parse_args()
n = args.envs
env = HipVecEnv(num_envs=n, lanes_per_walker=16, seed=1234)
env.reset_tensors(); env.debug_counters()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(args.steps, n, 8, device='cuda', generator=g), -1, 1)
cases = []
for t in range(args.steps):
    pre = env.get_state()
    env.step_tensors(acts[t])
    it, mx, nr, dv = env.debug_counters()
    hit = np.nonzero(mx >= 100)[0]
    if len(hit):
        cs = env.debug_capstate()
        for w in hit:
            cases.append(dict(t=t, w=int(w), q=cs[0:14, w].copy(), v=cs[16:30, w].copy(), warm=cs[32:46, w].copy(),
                              cur=pre['cursor'][:, w].copy(), a=acts[t, w].cpu().numpy().copy()))
print('cases with >= 100 iterations in one evaluation:', [(c['t'], c['w']) for c in cases])
m = 4
e32 = HipVecEnv(num_envs=m, lanes_per_walker=16, precision=32)
e32l = HipVecEnv(num_envs=m, lanes_per_walker=1, precision=32)
e64 = HipVecEnv(num_envs=m, lanes_per_walker=16, precision=64)
dump = {}
rep = lambda x: np.repeat(np.asarray(x, np.float64)[:, None], m, 1)
for ci, c in enumerate(cases[:8]):
    u = 300.0 * np.clip(c['a'].astype(np.float64), -1, 1)
    i_step = int(c['cur'][abi.DL_CUR_I_STEP])
    is_left = bool(env.refs.as_desc().step_is_left[i_step])
    if is_left:
        u = u[[4, 5, 6, 7, 0, 1, 2, 3]]; u[1] = -u[1]; u[5] = -u[5]
    U = rep(u)
    print(f'--- case t={c["t"]} walker={c["w"]} left={is_left} finite={np.isfinite(c["q"]).all() and np.isfinite(c["v"]).all() and np.isfinite(c["warm"]).all()}')
    print('    q', np.array2string(c['q'], precision=4)); print('    v', np.array2string(c['v'], precision=3)); print('    warm', np.array2string(c['warm'], precision=2))
    res = {}
    for name, e in (('f32g16', e32), ('f32l1', e32l), ('f64g16', e64)):
        e.set_state(qpos=rep(c['q']), qvel=rep(c['v']), warm=rep(c['warm']))
        qa, nc, ne, ni = e.forward(U)
        res[name] = qa[:, 0].astype(np.float64)
        print(f'    {name}: ncon={int(nc[0])} nefc={int(ne[0])} iters={int(ni[0])} |qacc|max={np.abs(qa[:,0]).max():.3e}')
    for k in ('q', 'v', 'warm'):
        dump[f'c{ci}_{k}'] = c[k]
    dump[f'c{ci}_u'] = u
    for k in res:
        dump[f'c{ci}_{k}'] = res[k]
np.savez(args.out, **dump)
print('saved', len(cases))
