import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'host_emu'))
import numpy as np
import emu as E
from oracle import oracle as O
from drloco_amd.vec_env import HipVecEnv
np.set_printoptions(precision=5, suppress=True, linewidth=200)
n = 1024
dev = HipVecEnv(num_envs=n, precision=32)
em = E.EmuEnv(dev.model, dev.refs, dev.cfg, n, 32)
rng = np.random.default_rng(3)
o1 = em.reset(); o2 = dev.reset()
print('reset obs diff', np.abs(o1 - o2).max())
s1, s2 = em.get_state(), dev.get_state()
print('warm diff', np.abs(s1['warm'] - s2['warm']).max(), 'cursor eq', np.array_equal(s1['cursor'], s2['cursor']))
for t in range(8):
    a = np.clip(0.5 * rng.standard_normal((n, 8)), -1, 1).astype(np.float32)
    qa, nc, ne, ni = em.forward(); qb, nc2, ne2, ni2 = dev.forward()
    print(t, 'fwd: ncon eq', np.array_equal(nc, nc2), 'niter emu/dev max', ni.max(), ni2.max(), 'qacc diff', np.abs(qa - qb).max())
    ob1, r1, d1, _, _ = em.step(a); ob2, r2, d2, _ = dev.step(a)
    bad = np.nonzero((d1.astype(bool) != d2) | (np.abs(r1 - r2) > 1e-3))[0]
    print(t, 'rew diff', np.abs(r1 - r2).max(), 'done mism', (d1.astype(bool) != d2).sum(), 'bad', bad[:10])
    if len(bad):
        i = bad[0]
        s1, s2 = em.get_state(), dev.get_state()
        print(' env', i, 'r', r1[i], r2[i], 'd', d1[i], d2[i], 'cur emu', s1['cursor'][:, i], 'dev', s2['cursor'][:, i])
        print(' q emu', s1['qpos'][:, i]); print(' q dev', s2['qpos'][:, i])
        break
