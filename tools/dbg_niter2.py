import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.vec_env import HipVecEnv
from oracle import oracle as O
np.set_printoptions(precision=6, suppress=True, linewidth=220)
env = HipVecEnv(num_envs=4096, reserved=1)
env0 = HipVecEnv(num_envs=4096, reserved=0)
env64 = HipVecEnv(num_envs=4096, reserved=1, precision=64)
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(400, 4096, 8, device='cuda', generator=g), -1, 1)
found = 0
for t in range(400):
    env.step_tensors(acts[t])
    st = env.get_state()
    qa, nc, ne, ni = env.forward()
    if ni.max() > 15:
        i = int(ni.argmax())
        env0.set_state(qpos=st['qpos'], qvel=st['qvel'], warm=st['warm'], cursor=st['cursor'], walked=st['walked'])
        qb, nc0, ne0, ni0 = env0.forward()
        orc = O.OracleEnv(env.model, env.refs, env.cfg, 1)
        orc.set_state(qpos=st['qpos'][:, i:i+1], qvel=st['qvel'][:, i:i+1], warm=st['warm'][:, i:i+1])
        qo, nco, neo, nio = orc.forward()
        env64.set_state(qpos=st['qpos'], qvel=st['qvel'], warm=st['warm'], cursor=st['cursor'], walked=st['walked'])
        q64, nc64, ne64, ni64 = env64.forward()
        print('  g16 f64 niter', ni64[i], 'max over all', ni64.max(), 'lane f32 max', ni0.max(), 'g16 f32 hist', np.bincount(ni)[:12], 'count>15', (ni > 15).sum())
        print('t', t, 'walker', i, 'niter g16', ni[i], 'lane', ni0[i], 'oracle', nio[0], 'nefc', ne[i], ne0[i], neo[0], 'ncon', nc[i])
        print(' qacc g16 ', qa[:, i]); print(' qacc lane', qb[:, i]); print(' qacc orc ', qo[:, 0])
        print(' q', st['qpos'][:, i]); print(' v', st['qvel'][:, i]); print(' warm', st['warm'][:, i])
        found += 1
        if found >= 2: break
print('done', found)
