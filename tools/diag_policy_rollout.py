#!/usr/bin/env python3
"""Section stamps of the policy phase INSIDE the persistent rollout kernel (workgroup 0, last control step; -DDL_EXP_POL_PROF build via DL_LIB_PATH)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
n, T = 4096, 64
names = ['start', 'obs staged + barrier', 'layer 1 done, h1 staged', 'after the h1 barrier', 'hidden layer done', 'heads done', 'after the heads barrier', 'end']
for moments in ('per_step', 'per_rollout'):
    venv = HipVecEnv(num_envs=n, seed=1234)
    vn = HipVecNormalize(venv); vn.reset()
    pol = HipPolicy(hidden=512, seed=99)
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
    last_obs, last_done = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
    acc = np.zeros((2, 8)); R = 6
    for i in range(R + 2):
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments=moments)
        torch.cuda.synchronize()
        out = (C.c_longlong * 16)()
        assert venv._lib.dl_debug_pol_prof(out) == 0
        s = np.array(list(out), dtype=np.float64).reshape(2, 8)
        if i >= 2: acc += s - s[0, 0]
    acc /= R
    print(moments)
    for k in range(8): print(f'  {names[k]:28s} wave 0: {acc[0, k]:8.0f}   wave 7: {acc[1, k]:8.0f}')
    venv.close()
