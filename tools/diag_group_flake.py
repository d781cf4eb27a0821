#!/usr/bin/env python3
"""tests/test_gpu_parity.py::test_env_group_handles_are_shards failed once in 13 runs (round 6 soak: handle 0's observations differ between the two-handle group and the shard run on
its own).  This loop repeats the test's two sides with the same seeds and compares EVERY repetition with the first one: which side moves (the group's concurrent streams or the
single handle), where the first difference is (step, walker, array) and how large.  usage: tools/diag_group_flake.py [repetitions] [chunk]"""
import sys
import numpy as np
import torch
from drloco_amd import models
from drloco_amd.group import HipEnvGroup
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T, H, n = 40, 2, 192
NAMES = ('observations', 'actions', 'values', 'log_probs', 'rewards', 'episode_starts')


def group_side():
    pol = HipPolicy(hidden=128, seed=6)
    grp = HipEnvGroup(T, num_envs=H * n, handles=H, seed=77, index_base=1000)
    grp.collect_rollouts(pol, chunk=chunk)
    grp.join()
    torch.cuda.synchronize()
    out = [{k: getattr(grp.bufs[h], k).clone().cpu().numpy() for k in NAMES} for h in range(H)]
    grp.close()
    return pol, out


def single_side(pol):
    out = []
    for h in range(H):
        vn = HipVecNormalize(HipVecEnv(num_envs=n, seed=77, env_index_base=1000 + h * n))
        p2 = HipPolicy(hidden=128, seed=6, index_base=1000 + h * n)
        p2.load_state(pol.w1, pol.b1, pol.w2, pol.b2, pol.wa, pol.ba, pol.wv, pol.bv, pol.log_std)
        buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
        vn.reset()
        last_obs = vn.norm_obs_t.clone(); last_done = torch.ones(n, dtype=torch.uint8, device='cuda')
        buf.collect_rollouts(vn, p2, last_obs, last_done, persistent=False)
        torch.cuda.synchronize()
        out.append({k: getattr(buf, k).clone().cpu().numpy() for k in NAMES})
        vn.venv.close()
    return out


def first_diff(a, b):
    """(step, array, walkers that differ at that step, max |difference| there) of the first step at which any array differs"""
    best = None
    for k in NAMES:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        d = (x != y)
        if d.any():
            t = int(np.argwhere(d.reshape(d.shape[0], -1).any(axis=1))[0, 0])
            w = np.argwhere(d[t].reshape(n, -1).any(axis=1))[:, 0]
            item = (t, k, w.tolist()[:8], len(w), float(np.abs(x[t] - y[t]).max()))
            if best is None or (t, NAMES.index(k)) < (best[0], NAMES.index(best[1])):
                best = item
    return best


pol0, g0 = group_side()
s0 = single_side(pol0)
print('repetition 0: group vs single', [first_diff(g0[h], s0[h]) for h in range(H)], flush=True)
bad = 0
for r in range(1, reps):
    pol, g = group_side()
    s = single_side(pol)
    for h in range(H):
        for side, x, ref in (('group', g[h], g0[h]), ('single', s[h], s0[h])):
            d = first_diff(x, ref)
            if d is not None:
                bad += 1
                print(f'repetition {r} handle {h}: the {side} side differs from repetition 0: first at step {d[0]} in {d[1]}, {d[3]} walkers {d[2]}, max |diff| {d[4]:.3e}', flush=True)
print(f'{bad} deviating (repetition, handle, side) of {reps - 1} repetitions, chunk {chunk}')
