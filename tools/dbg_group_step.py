import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from drloco_amd.vec_env import HipVecEnv
np.set_printoptions(precision=5, suppress=True, linewidth=220)
n = 256
dev = HipVecEnv(num_envs=n, precision=64, reserved=1)
orc = O.OracleEnv(dev.model, dev.refs, dev.cfg, n)
o1 = orc.reset(); o2 = dev.reset()
print('reset diff', np.abs(o1 - o2).max())
rng = np.random.default_rng(1)
nd = 0
for t in range(120):
    a = np.clip(0.5 * rng.standard_normal((n, 8)), -1, 1).astype(np.float32)
    ob1, r1, d1, t1, _ = orc.step(a.astype(np.float64)); ob2, r2, d2, infos = dev.step(a)
    dm = (d1.astype(bool) != d2).sum()
    oe = np.abs(ob1 - ob2).max(); re = np.abs(r1 - r2).max()
    nd += d2.sum()
    if t % 20 == 0 or dm or oe > 1e-4: print(t, 'done mism', dm, 'obs err', oe, 'rew err', re)
    if dm or oe > 1e-3: break
s1, s2 = orc.get_state(), dev.get_state()
print('dones', nd, 'cursor eq', np.array_equal(s1['cursor'], s2['cursor']), 'walked err', np.abs(s1['walked'] - s2['walked']).max())
for name in ('ep_len_smoothed', 'ep_ret_smoothed', 'mean_reward_smoothed'):
    print(name, np.abs(np.array(dev.get_attr(name)) - orc.stats(name)).max())
# timing f32, 4096 walkers
for variant in (0, 1):
    env = HipVecEnv(num_envs=4096, reserved=variant)
    env.reset_tensors()
    g = torch.Generator(device='cuda'); g.manual_seed(4321)
    acts = torch.clamp(0.5 * torch.randn(160, 4096, 8, device='cuda', generator=g), -1, 1)
    for t in range(60): env.step_tensors(acts[t])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(60, 160): env.step_tensors(acts[t])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print('variant', variant, 'ms/step', dt * 1e3, 'env-steps/s', 4096 / dt, 'mean rew', env.rew.mean().item())
