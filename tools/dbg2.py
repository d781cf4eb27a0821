import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drloco_amd.vec_env import HipVecEnv
np.set_printoptions(precision=5, suppress=True, linewidth=200)
for prec in (32, 64):
    n = 1024
    dev = HipVecEnv(num_envs=n, precision=prec)
    o = dev.reset()
    s = dev.get_state()
    print(prec, 'obs nan rows', np.isnan(o).any(1).sum(), 'q nan', np.isnan(s['qpos']).any(0).sum(), 'v nan', np.isnan(s['qvel']).any(0).sum(), 'warm nan', np.isnan(s['warm']).any(0).sum())
    bad = np.nonzero(np.isnan(o).any(1))[0]
    print(' bad envs', bad[:20])
    if len(bad):
        i = bad[0]
        print(' obs', o[i]); print(' q', s['qpos'][:, i]); print(' cur', s['cursor'][:, i])
    qacc, nc, ne, ni = dev.forward()
    print(' forward: niter hist', np.bincount(ni), 'nan qacc envs', np.isnan(qacc).any(0).sum(), 'ncon', np.bincount(nc))
    badf = np.nonzero(np.isnan(qacc).any(0) | (ni >= 50))[0]
    print(' bad forward envs', badf[:20], 'lanes', badf[:20] % 64)
