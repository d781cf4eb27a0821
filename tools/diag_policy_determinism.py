#!/usr/bin/env python3
"""Are the policy-in-the-loop rollouts repeatable?  The same 4096-walker x T-step rollout (fresh handles, same seeds) R times in each form -- the persistent kernel with exact
per-step moments, the pair-by-pair kernel with per-rollout moments, three launches per control step -- hashed: every repetition of a form must give the bits of its first, and
the exact persistent form the bits of the launch form.  usage: tools/diag_policy_determinism.py [T] [R] [walker: straight | loco3d]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd import models
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
LOCO3D = len(sys.argv) > 3 and sys.argv[3] == 'loco3d'
n = 4096
kw = {}
if LOCO3D:
    from drloco_amd import mocap
    ang, vel = mocap.synthetic_loco3d(L=60000, seed=0)
    kw = dict(env_id=models.WALKER_165CM, refs=mocap.loco3d_table(ang, vel))


def run(form):
    venv = HipVecEnv(kw.get('env_id', models.STRAIGHT_WALKER), num_envs=n, seed=1234, lanes_per_walker='split', **({'refs': kw['refs']} if LOCO3D else {}))
    vn = HipVecNormalize(venv); vn.blocked_reduce = True          # the launch form's moment reduction in the persistent kernel's order (what tests/test_gpu_persistent.py compares)
    pol = HipPolicy(obs_dim=venv.obs_dim, act_dim=venv.nu, hidden=512, seed=5)
    buf = HipRolloutBuffer(T, n, venv.obs_dim, venv.nu, torch.device('cuda'))
    vn.reset()
    last_obs = vn.norm_obs_t.clone(); last_done = torch.ones(n, dtype=torch.uint8, device='cuda')
    if form == 'exact':
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
    elif form == 'per_rollout':
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments='per_rollout')
    else:
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=False)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for name in ('observations', 'actions', 'values', 'log_probs', 'rewards', 'episode_starts'):
        h.update(getattr(buf, name).cpu().numpy().tobytes())
    h.update(last_obs.cpu().numpy().tobytes())
    venv.close()
    return h.hexdigest()[:16]


first = {}
for form in ('exact', 'per_rollout', 'launches'):
    bad = 0
    for r in range(R if form != 'launches' else max(2, R // 4)):
        d = run(form)
        first.setdefault(form, d)
        if d != first[form]:
            bad += 1
            print(f'{form}: repetition {r} DIFFERS ({d} against {first[form]})', flush=True)
    print(f'{form}: {bad} repetitions differ from the first ({first[form]})', flush=True)
print('exact persistent form == launch form:', first['exact'] == first['launches'])
