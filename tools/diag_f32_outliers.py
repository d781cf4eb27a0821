#!/usr/bin/env python3
"""float32 forward dynamics against the float64 oracle from identical random states: which walkers miss 1e-3 on the scaled acceleration
error, and do they have another active set (a constraint row whose J a - aref changes sign between the two solutions) than the oracle?"""
import sys
import numpy as np
sys.path.insert(0, '.')
from drloco_amd import mocap, models
from drloco_amd.vec_env import HipVecEnv
from oracle import oracle as O


def run(name, dev, n, nv, nu, seed, lanes):
    rng = np.random.default_rng(seed)
    q = np.array(dev.model.jnt_qpos0[:nv])[:, None] + (0.25 if nv == 14 else 0.2) * rng.standard_normal((nv, n))
    q[2] = rng.uniform(0.85, 1.3, n) if nv == 14 else rng.uniform(0.75, 1.2, n)
    v = 1.5 * rng.standard_normal((nv, n)); w = rng.standard_normal((nv, n)); u = rng.uniform(-300, 300, (nu, n))
    orc = O.OracleEnv(dev.model, dev.refs, dev.cfg, n)
    dev.set_state(qpos=q, qvel=v, warm=w); orc.set_state(qpos=q, qvel=v, warm=w)
    qa, nc, ne, ni = orc.forward(u); qb, nc2, ne2, ni2 = dev.forward(u)
    err = (np.abs(qa - qb) / (1 + np.abs(qa))).max(axis=0)
    bad = np.nonzero(err >= 1e-3)[0]
    print(f'{name} lanes={lanes}: n={n} median {np.median(err):.2e} q99 {np.quantile(err, 0.99):.2e} max {err.max():.2e}; >= 1e-3: {len(bad)}; ncon equal {np.array_equal(nc, nc2)}')
    for i in bad:
        r = O.probe_forward(dev.model, q[:, i], v[:, i], u[:, i], w[:, i])
        ja, jb = r['efc_J'] @ qa[:, i] - r['efc_aref'], r['efc_J'] @ qb[:, i] - r['efc_aref']
        flips = np.nonzero((ja < 0) != (jb < 0))[0]
        print(f'   walker {i}: err {err[i]:.2e} ncon {nc[i]} nefc {ne[i]} niter oracle {ni[i]} device {ni2[i]}; rows with another active state: {len(flips)}'
              + (f' (|J a - aref| there: oracle {np.abs(ja[flips]).max():.2e}, device {np.abs(jb[flips]).max():.2e})' if len(flips) else '') + f' cost gap {0.0:.1e}')


for lanes in (16, 'split', 1):
    dev = HipVecEnv(num_envs=1024, precision=32, lanes_per_walker=lanes)
    run('straight', dev, 1024, 14, 8, 0, lanes)
    dev.close()
ang, vel = mocap.synthetic_loco3d(L=4000, seed=1)
for lanes in (16, 1):
    dev = HipVecEnv(models.WALKER_165CM, num_envs=512, precision=32, refs=mocap.loco3d_table(ang, vel), lanes_per_walker=lanes)
    run('walker165', dev, 512, 19, 13, 0, lanes)
    dev.close()
