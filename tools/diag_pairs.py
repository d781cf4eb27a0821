#!/usr/bin/env python3
"""The per-pair per-rollout kernel (-DDL_EXP_ROLLOUT_PAIRS build via DL_LIB_PATH) against dl_policy_forward on the recorded observations: which rows of which
steps leave the policy phase with other outputs (EXPERIMENTS.md, round 4: always the third walker of a pair, only while other pairs run).  ONLY0=1 with a
-DDL_EXP_PAIR_ONLY0 build: one pair per workgroup active."""
import sys, os
sys.path.insert(0, os.getcwd())
import ctypes as C, numpy as np, torch
from drloco_amd import lib as L
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize, _ptr, _stream
n, T = 1000, 24
venv = HipVecEnv(num_envs=n, seed=21); venv.set_split(True)
vn = HipVecNormalize(venv); vn.blocked_reduce = True; vn.reset()
pol = HipPolicy(hidden=512, seed=4)
buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
lo, ld = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
buf.collect_rollouts(vn, pol, lo, ld, persistent=True)
c0 = pol.counter
try:
    buf.collect_rollouts(vn, pol, lo, ld, persistent=True, moments="per_rollout")
except Exception as ex:
    print("FAULT", str(ex)[:300])
torch.cuda.synchronize()
p2 = HipPolicy(hidden=512, seed=4)
for t in range(T):
    obs = buf.observations[t].contiguous()
    p2.counter = c0 + t
    a, v, lp = p2.forward(obs)
    a2, v2, l2 = torch.empty_like(a), torch.empty_like(v), torch.empty_like(lp)
    p = p2._params()
    L.check(p2._lib.dl_policy_forward_pair(C.byref(p), _ptr(p2._packed_weights()), _ptr(obs), n, None, p2.seed, c0 + t, p2.index_base, 0, _ptr(a2), _ptr(v2), _ptr(l2), _stream()))
    torch.cuda.synchronize()
    bad = (a != buf.actions[t]).any(1) | (v != buf.values[t])
    if os.environ.get("ONLY0"): bad &= ((torch.arange(n, device="cuda") // 4) % 4 == 0)
    print(t, 'rows==pair standalone:', bool(torch.equal(a, a2) and torch.equal(v, v2)), '| kernel==rows:', bool(torch.equal(a, buf.actions[t]) and torch.equal(v, buf.values[t])),
          '| rows differing', int(bad.sum()), bad.nonzero()[:8, 0].tolist(), 'max |da|', float((a - buf.actions[t]).abs().max()), 'max |dv|', float((v - buf.values[t]).abs().max()))
    for r in bad.nonzero()[:3, 0].tolist():
        print('   row', r, 'start flag', int(buf.episode_starts[t][r]), 'prev start', int(buf.episode_starts[t-1][r]) if t else -1, 'da', (buf.actions[t][r] - a[r]).cpu().numpy().round(4), 'dv', float(buf.values[t][r] - v[r]), 'dlp', float(buf.log_probs[t][r] - lp[r]),
              '| neighbours start flags', buf.episode_starts[t][r - 2:r + 2].cpu().numpy())
    if t and bad.any():
        p2.counter = c0 + t
        ap, vp, lpp = p2.forward(buf.observations[t - 1].contiguous())
        for r in bad.nonzero()[:3, 0].tolist():
            print('   row', r, 'kernel output == policy of the PREVIOUS step observation:', bool(torch.equal(ap[r], buf.actions[t][r])), float((ap[r] - buf.actions[t][r]).abs().max()), 'value', float(vp[r] - buf.values[t][r]))
