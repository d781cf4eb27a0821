#!/usr/bin/env python3
"""Throughput of the drop-in VecEnv surface -- numpy in / numpy out, what an UNCHANGED drloco/train.py drives through SB3's collect_rollouts
(drloco/train.py:110-133 -> env.step(numpy actions)): HipVecNormalize.step(actions) per control step, host -> device copy of the actions, the step's launches,
device -> host copies of obs / reward / done (+ the terminal observations of finished walkers) and the info dicts included.  Where the time goes: the same loop with
the pieces taken apart (device work alone = step_tensors + one synchronize; copies; info dicts fresh vs reused).
usage: python3 tools/bench_vecenv_api.py [walkers ...]   (default 8 256 4096)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from drloco_amd.vec_env import vec_env

sizes = [int(x) for x in sys.argv[1:]] or [8, 256, 4096]
K = 300
print(f'{"walkers":>8s} {"us / VecEnv.step":>17s} {"M env-steps/s":>14s} | {"fresh info dicts":>17s} {"random actions: episodes end":>29s} | {"device work alone":>18s} {"3 D2H copies":>13s} {"N fresh dicts":>14s}')
for n in sizes:
    env = vec_env(num_envs=n, seed=1)
    env.reset()
    rng = np.random.default_rng(0)
    acts = np.clip(0.5 * rng.standard_normal((50, n, 8)), -1, 1).astype(np.float32)
    zero = np.zeros((n, 8), np.float32)          # zero torques: the walkers stand for a few hundred steps -> hardly any terminal_observation traffic

    def loop(a_of_t, k=K):
        for t in range(20):
            env.step(a_of_t(t))
        t0 = time.perf_counter()
        for t in range(k):
            obs, rew, done, infos = env.step(a_of_t(t))
        return (time.perf_counter() - t0) / k * 1e6
    env.reset()
    us = loop(lambda t: zero)
    env.venv.reuse_infos = False
    env.reset()
    us_fresh = loop(lambda t: zero)
    env.venv.reuse_infos = True
    env.reset()
    us_rand = loop(lambda t: acts[t % 50])          # +-300 N m noise: walkers fall all the time (terminal observations, ep_lens bookkeeping)
    # the pieces
    a_dev = torch.as_tensor(zero, device=env.venv.device)
    env.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        env.step_tensors(a_dev); torch.cuda.synchronize()
    us_dev = (time.perf_counter() - t0) / K * 1e6
    o, r, d = env.norm_obs_t, env.norm_rew_t if hasattr(env, 'norm_rew_t') else env.venv.rew, env.venv.done
    t0 = time.perf_counter()
    for t in range(K):
        o.cpu().numpy(); r.cpu().numpy(); d.cpu().numpy()
    us_copy = (time.perf_counter() - t0) / K * 1e6
    t0 = time.perf_counter()
    for t in range(K):
        [{} for _ in range(n)]
    us_dicts = (time.perf_counter() - t0) / K * 1e6
    print(f'{n:8d} {us:17.1f} {n / us:14.3f} | {us_fresh:17.1f} {us_rand:29.1f} | {us_dev:18.1f} {us_copy:13.1f} {us_dicts:14.1f}', flush=True)
    env.close()
