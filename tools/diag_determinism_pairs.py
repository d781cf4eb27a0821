#!/usr/bin/env python3
"""The per-rollout (pair-by-pair) rollout kernel repeated from the same start: every repetition must give the same bits although the pairs run asynchronously."""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import torch
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
for n, T, R in ((4096, 128, 40), (9000, 48, 25)):
    first, nd = None, 0
    for rep in range(R):
        venv = HipVecEnv(num_envs=n, seed=1234)
        vn = HipVecNormalize(venv); vn.reset()
        pol = HipPolicy(hidden=512, seed=99)
        buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
        lo, ld = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
        buf.collect_rollouts(vn, pol, lo, ld, persistent=True)                       # moments with some history
        buf.collect_rollouts(vn, pol, lo, ld, persistent=True, moments='per_rollout')
        torch.cuda.synchronize()
        h = hashlib.sha1(b''.join(x.cpu().numpy().tobytes() for x in (buf.observations, buf.actions, buf.rewards, buf.values, buf.log_probs, buf.episode_starts, lo)) + vn.obs_rms.mean.tobytes() + vn.obs_rms.var.tobytes()).hexdigest()[:16]
        if first is None: first = h
        elif h != first: nd += 1
        venv.close()
    print(f'per-rollout kernel (pairs), {n} walkers x {T} steps: {nd} of {R - 1} repetitions differ from the first ({first})')
