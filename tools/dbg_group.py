import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from drloco_amd.vec_env import HipVecEnv
np.set_printoptions(precision=5, suppress=True, linewidth=220)
n = 256
for prec in (64, 32):
    dev = HipVecEnv(num_envs=n, precision=prec, reserved=1)
    orc = O.OracleEnv(dev.model, dev.refs, dev.cfg, n)
    rng = np.random.default_rng(0)
    for case in ('free', 'contact'):
        q = np.array(dev.model.jnt_qpos0[:14])[:, None] + 0.25 * rng.standard_normal((14, n))
        q[2] = 3.0 if case == 'free' else rng.uniform(0.85, 1.3, n)
        if case == 'free':
            q[6:] = np.clip(q[6:], [[-0.8], [-0.7], [0.05], [-0.3], [-0.8], [-0.05], [0.05], [-0.3]], [[0.8], [0.05], [2.5], [0.6], [0.8], [0.7], [2.5], [0.6]])
        v = 1.5 * rng.standard_normal((14, n)); w = rng.standard_normal((14, n)); u = rng.uniform(-300, 300, (8, n))
        dev.set_state(qpos=q, qvel=v, warm=w); orc.set_state(qpos=q, qvel=v, warm=w)
        qa, nc, ne, ni = orc.forward(u); qb, nc2, ne2, ni2 = dev.forward(u)
        err = np.abs(qa - qb) / (1 + np.abs(qa))
        print(prec, case, 'ncon eq', np.array_equal(nc, nc2), 'nefc eq', np.array_equal(ne, ne2), 'niter eq', (ni == ni2).mean(), 'max nefc', ne.max(), 'qacc err max', err.max(), 'median', np.median(err.max(0)), 'nan', np.isnan(qb).sum())
        if err.max() > 1e-6 and prec == 64:
            i = int(err.max(0).argmax())
            print(' worst env', i, 'ncon', nc[i], nc2[i], 'nefc', ne[i], ne2[i], 'niter', ni[i], ni2[i])
            print(' oracle', qa[:, i]); print(' device', qb[:, i])
