#!/usr/bin/env python3
"""Count the wrong policy rows of the per-rollout kernel (k_rollout_pairs) of whatever library DL_LIB_PATH names: every recorded action / value of R rollouts of
T steps x 4096 walkers is recomputed from its recorded observation with dl_policy_forward and compared bit for bit.  The product library gives 0; the variants of
tools/asm_bisect.py give the round-4 defect or not (EXPERIMENTS.md, "the 4x4x1 defect, found").  GPU box; POLP_R = number of rollouts (default 4 = 524 288 rows), POLP_HIDDEN = 512 / 256 / 128."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

n, T, R = 4096, 32, int(os.environ.get('POLP_R', '4'))
HID = int(os.environ.get('POLP_HIDDEN', '512'))
venv = HipVecEnv(num_envs=n, seed=21)
venv.set_split(True)
vn = HipVecNormalize(venv)
vn.blocked_reduce = True
vn.reset()
pol = HipPolicy(hidden=HID, seed=4)
buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
lo, ld = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
p2 = HipPolicy(hidden=HID, seed=4)
bad = rows = 0
byrow = [0, 0, 0, 0]
by_t = [0] * T          # where in the rollout (the pairs of a workgroup start their first policy phase together and drift apart afterwards)
by_pair = [0, 0, 0, 0]          # which pair of its workgroup
for r in range(R):
    c0 = pol.counter
    try:
        buf.collect_rollouts(vn, pol, lo, ld, persistent=True, moments='per_rollout')
    except Exception as ex:
        print('FAULT', str(ex)[:200])
        break
    torch.cuda.synchronize()
    for t in range(T):
        p2.counter = c0 + t
        a, v, lp = p2.forward(buf.observations[t])
        w = (a != buf.actions[t]).any(1) | (v != buf.values[t])
        bad += int(w.sum())
        rows += n
        idx = w.nonzero()[:, 0]
        for k in range(4):
            byrow[k] += int((idx % 4 == k).sum())
            by_pair[k] += int(((idx // 4) % 4 == k).sum())
        by_t[t] += int(w.sum())
print(os.environ.get('DL_LIB_PATH', 'product').split('/')[-1], 'hidden', HID, ': wrong rows', bad, 'of', rows, '; by row of the pair', byrow)
if bad:
    print('   by control step of the rollout:', by_t)
    print('   by pair of the workgroup:', by_pair)
