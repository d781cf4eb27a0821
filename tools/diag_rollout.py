#!/usr/bin/env python3
"""Where a control step of the persistent rollout kernel goes (needs a build with -DDL_EXP_ROLLOUT_PROF, selected with DL_LIB_PATH):
per workgroup the shader-clock cycles in the policy phase, the env phase, the moment sums + exchange, and waiting in the exchange."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from drloco_amd import lib
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

n, T = 4096, 256
for moments in ('per_step', 'per_rollout'):
    venv = HipVecEnv(num_envs=n, seed=1234)
    vn = HipVecNormalize(venv); vn.reset()
    pol = HipPolicy(hidden=512, seed=99)
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
    last_obs, last_done = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
    for _ in range(3):
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments=moments)
    torch.cuda.synchronize()
    prof = torch.zeros(n // 16 * 4 * 11 + 512 * (n // 16) * 4, dtype=torch.int64, device='cuda')          # (+ the per-step records of the PROF builds since round 5)
    lib.check(venv._lib.dl_debug_rollout_prof(venv._h, C.c_void_p(prof.data_ptr()), None))
    sec = prof[n // 16 * 4:n // 16 * 4 * 11].view(10, n // 4).cpu().numpy().astype(np.float64)
    p = prof[:n // 16 * 4].view(n // 16, 4).cpu().numpy().astype(np.float64) / T
    tot = p[:, :3].sum(1)
    print(f'{moments}: cycles per control step and workgroup (mean / min / max over {n // 16} workgroups; shader clock)')
    for k, name in enumerate(('policy phase', 'env phase', 'sums + exchange', '  of which waiting')):
        print(f'  {name:20s} {p[:, k].mean():10.0f} {p[:, k].min():10.0f} {p[:, k].max():10.0f}')
    print(f'  {"total":20s} {tot.mean():10.0f} {tot.min():10.0f} {tot.max():10.0f}')
    names = ('smooth dynamics', 'constraints', 'rows / J^T f / Hessian', 'factor + solve', 'J dir / M dir', 'line search + step', '#iterations', 'whole env step', 'before the physics', 'after the physics')
    print('  last control step, per dynamics wave (mean / max over %d waves):' % sec.shape[1])
    for k, name in enumerate(names):
        print(f'    {name:26s} {sec[k].mean():10.0f} {sec[k].max():10.0f}')
    wg = sec[7].reshape(-1, 4)
    print(f'    max over the 4 waves of a workgroup: mean {wg.max(1).mean():.0f}; mean wave {sec[7].mean():.0f}')
    venv.close()
