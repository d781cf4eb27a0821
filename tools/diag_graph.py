#!/usr/bin/env python3
"""Does a HIP graph over the launches of dl_rollout_policy shorten the gaps between them?  Captures one whole rollout (T x 3 launches) with
torch.cuda.CUDAGraph and times replays against direct calls (the captured policy noise counter is frozen: a timing experiment only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

n, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 512
venv = HipVecEnv(num_envs=n, seed=1234)
vn = HipVecNormalize(venv)
buf = HipRolloutBuffer(T, n, venv.obs_dim, venv.nu, torch.device('cuda'), gamma=0.995, gae_lambda=0.95)
policy = HipPolicy(obs_dim=venv.obs_dim, act_dim=venv.nu, hidden=512, seed=99)
vn.reset()
last_obs, last_done = vn.norm_obs_t, buf.next_starts
last_done.fill_(1)


def direct():
    buf.collect_rollouts(vn, policy, last_obs, last_done, persistent=False)        # the launch-per-step form is what a graph would shorten


def timeit(f, k=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    td = timeit(direct)
    print(f'direct: {td * 1e3:.2f} ms per rollout, {td / T * 1e6:.1f} us per control step, {n * T / td / 1e6:.2f} M env-steps/s')
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        direct()
    tg = timeit(g.replay)
    print(f'graph : {tg * 1e3:.2f} ms per rollout, {tg / T * 1e6:.1f} us per control step, {n * T / tg / 1e6:.2f} M env-steps/s')
