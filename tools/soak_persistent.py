#!/usr/bin/env python3
"""Long run of the persistent rollout kernel at several sizes (one / eight / five blocks per workgroup, exact and per-rollout moments): fault word clear, everything finite.
usage (GPU box): python3 tools/soak_persistent.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from drloco_amd import lib as L
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
for n, T, R, moments in ((32768, 64, 40, 'per_step'), (4096, 256, 60, 'per_step'), (20000, 64, 30, 'per_rollout')):
    venv = HipVecEnv(num_envs=n, seed=5)
    vn = HipVecNormalize(venv); vn.reset()
    pol = HipPolicy(hidden=512, seed=3)
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
    lo, ld = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
    t0 = time.time(); eps = 0
    for r in range(R):
        buf.collect_rollouts(vn, pol, lo, ld, persistent=True, moments=moments)
        buf.compute_returns_and_advantage(buf.values[-1], ld)
        eps += int(buf.episode_starts.sum())
        assert torch.isfinite(buf.observations).all() and torch.isfinite(buf.rewards).all() and torch.isfinite(buf.advantages).all()
    torch.cuda.synchronize()
    L.check(venv._lib.dl_fault_check(venv._h, None))
    print(f'{n} walkers x {T} steps x {R} rollouts ({moments}): ok, {eps} episode starts, form {buf.last_form}, {n*T*R/(time.time()-t0)/1e6:.1f} M env-steps/s incl. host checks, obs var max {float(vn.obs_rms.var.max()):.3g}')
    venv.close()
