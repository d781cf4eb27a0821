#!/usr/bin/env python3
"""Why does the batched deterministic evaluation (drloco_amd/evaluation.py) walk less than the training episodes?  Trains for a few million
steps (examples/train_ppo.py, saved and re-loaded through drloco_amd/checkpoint.py), then evaluates (a) as eval_walking does (deterministic
init states, mean action), (b) from RSI init states with the mean action, (c) from the deterministic init states with sampled actions, and
splits (a) by the parity of the evaluation counter k (quirk Q3: every evaluation episode reads reference step 0 while is_step_left follows k)."""
import glob, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'examples'))
import numpy as np
import torch
import train_ppo
from drloco_amd import checkpoint
from drloco_amd.evaluation import evaluate_walking, evaluate_walking_host_loop, make_eval_env
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

d = tempfile.mkdtemp()
hist, pol_mem, vn_mem = train_ppo.train(mio=float(sys.argv[1]) if len(sys.argv) > 1 else 6.0, seed=1, quiet=True, save_path=d, return_objects=True)
print('training: ep_len %.0f, walked %.1f m' % (hist[-1]['ep_len'], hist[-1]['moved_distance']))
pol = checkpoint.load_policy_zip(glob.glob(os.path.join(d, 'models', '*.zip'))[0], seed=5)
train_vn = HipVecNormalize.load(glob.glob(os.path.join(d, 'envs', 'env_*'))[0], HipVecEnv(num_envs=20, seed=1))


class Sampled:          # stochastic actions instead of the mean
    def __init__(self, p): self.p = p
    def forward(self, obs, deterministic=True): return self.p.forward(obs, deterministic=False)


def show(name, res):
    dist, dur = np.array(res['moved_distances']), np.array(res['ep_durs'])
    print(f'{name:58s} mean distance {dist.mean():5.1f} m, mean episode length {dur.mean():6.0f}, stable walks {res["count_stable_walks"]}/20;  even k {dist[0::2].mean():5.1f} m / odd k {dist[1::2].mean():5.1f} m')


show('(a) deterministic init states, mean action (eval_walking)', evaluate_walking(make_eval_env(train_vn), pol))
ev = make_eval_env(train_vn)
ev.venv.activate_evaluation(False)
show('(b) RSI init states, mean action', evaluate_walking(ev, pol))
show('(c) deterministic init states, sampled actions', evaluate_walking_host_loop(make_eval_env(train_vn), Sampled(pol)))
ev = make_eval_env(train_vn)
ev.venv.activate_evaluation(False)
show('(d) RSI init states, sampled actions (= training episodes)', evaluate_walking_host_loop(ev, Sampled(pol)))
ev = make_eval_env(vn_mem)
ev.venv.activate_evaluation(False)
show('(e) as (d) with the in-memory policy and moments', evaluate_walking_host_loop(ev, Sampled(pol_mem)))
show('(g) eval_walking with the training walkers\' step counter (history=training)', evaluate_walking(make_eval_env(vn_mem, history='training'), pol_mem))
for k in ('w1', 'b1', 'w2', 'b2', 'wa', 'ba', 'wv', 'bv', 'log_std'):
    assert torch.equal(getattr(pol, k), getattr(pol_mem, k).detach()), k
print('checkpoint round trip of the policy: identical tensors; obs mean max diff', np.abs(train_vn.obs_rms.mean - vn_mem.obs_rms.mean).max(), 'var', np.abs(train_vn.obs_rms.var - vn_mem.obs_rms.var).max())
# (f) the training env itself, continued for one more rollout with the sampled policy: episode lengths of the episodes that END in it
from drloco_amd.rollout import HipRolloutBuffer
buf = HipRolloutBuffer(3000, vn_mem.num_envs, 29, 8, torch.device('cuda'))
last_obs = vn_mem.norm_obs_t.clone(); last_done = torch.zeros(vn_mem.num_envs, dtype=torch.uint8, device='cuda')
vn_mem.training = False
buf.collect_rollouts(vn_mem, pol_mem, last_obs, last_done)
torch.cuda.synchronize()
print('(f) training env continued for 3000 steps with frozen moments: episodes ended', int(buf.episode_starts[1:].sum()), 'of', vn_mem.num_envs, 'walkers; ep_len_smoothed', np.mean(vn_mem.get_attr('ep_len_smoothed')))
