#!/usr/bin/env python3
"""Is the split-workgroup step kernel deterministic?  The same 4096-walker x T-step rollout R times as ONE multi-step launch and R times as T single-step launches:
every repetition of a form must give the same bits, and the two forms the same bits as each other.  usage: tools/diag_determinism.py [T] [R]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.vec_env import HipVecEnv
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 4096
LOCO3D = os.environ.get('DET_WALKER') == 'loco3d'          # the 19-dof walker's split workgroups (round 5)
NU, OBS = (13, 47) if LOCO3D else (8, 29)
if LOCO3D:
    from drloco_amd import mocap, models
    _ang, _vel = mocap.synthetic_loco3d(L=60000, seed=0)
    _table = mocap.loco3d_table(_ang, _vel)
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(T, n, NU, device='cuda', generator=g), -1, 1)
def run(multi):
    env = HipVecEnv(models.WALKER_165CM, num_envs=n, seed=1234, refs=_table, lanes_per_walker='split') if LOCO3D else HipVecEnv(num_envs=n, seed=1234, lanes_per_walker='split')
    env.reset_tensors()
    if multi:
        o, r, d = env.rollout_fixed(acts)
    else:
        o = torch.zeros(T, n, OBS, device='cuda'); r = torch.zeros(T, n, device='cuda'); d = torch.zeros(T, n, dtype=torch.uint8, device='cuda')
        for t in range(T):
            env.step_tensors(acts[t], obs_out=o[t], rew_out=r[t], done_out=d[t])
    torch.cuda.synchronize()
    o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
    env.close()
    return o, r, d
ref = {}
FORMS = (True,) if os.environ.get('DET_MULTI_ONLY') else (True, False)
for multi in FORMS:
    for rep in range(R):
        o, r, d = run(multi)
        key = 'multi' if multi else 'single'
        if key not in ref:
            ref[key] = (o, r, d)
            print(key, 'reference run: episodes ended', int(d.sum()), 'sha', hashlib.sha1(o.tobytes() + r.tobytes() + d.tobytes()).hexdigest()[:12], flush=True)
            continue
        o0, r0, d0 = ref[key]
        if np.array_equal(o, o0) and np.array_equal(r, r0) and np.array_equal(d, d0):
            print(key, rep, 'identical', flush=True)
        else:
            bad = np.nonzero((o != o0).any(axis=2) | (r != r0))
            t0 = bad[0].min(); ws = np.unique(bad[1][bad[0] == t0])
            print(key, rep, 'DIFFERS: first at step', t0, 'walkers', ws[:8], '(wave %s, workgroup %s)' % (np.unique(ws // 4)[:4], np.unique(ws // 16)[:4]), 'max |d obs| there', np.abs(o[t0] - o0[t0]).max(), 'reward diff', np.abs(r[t0] - r0[t0]).max(), flush=True)
if len(FORMS) == 2:
    a, b = ref['multi'], ref['single']
    print('multi == single:', all(np.array_equal(x, y) for x, y in zip(a, b)))

# ---- the persistent rollout kernel (policy in the loop): DET_PERSISTENT=<walkers> repeats one rollout of T steps R times from the same start
if os.environ.get('DET_PERSISTENT'):
    from drloco_amd.policy import HipPolicy
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecNormalize
    npers = int(os.environ['DET_PERSISTENT'])
    first = None
    ndiff = 0
    for rep in range(R):
        venv = HipVecEnv(num_envs=npers, seed=1234)
        vn = HipVecNormalize(venv); vn.blocked_reduce = True; vn.reset()
        pol = HipPolicy(hidden=512, seed=99)
        buf = HipRolloutBuffer(T, npers, 29, 8, torch.device('cuda'))
        last_obs, last_done = vn.norm_obs_t.clone(), torch.ones(npers, dtype=torch.uint8, device='cuda')
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
        torch.cuda.synchronize()
        h = hashlib.sha1(b''.join(x.cpu().numpy().tobytes() for x in (buf.observations, buf.actions, buf.rewards, buf.values, buf.log_probs, buf.episode_starts, last_obs))).hexdigest()[:16]
        if first is None: first = h
        elif h != first: ndiff += 1; print('persistent', rep, 'DIFFERS', h, flush=True)
        venv.close()
    print(f'persistent kernel, {npers} walkers x {T} steps: {ndiff} of {R - 1} repetitions differ from the first ({first})')
