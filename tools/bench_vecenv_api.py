#!/usr/bin/env python3
"""Throughput of the drop-in VecEnv surface (numpy in / numpy out, what an unmodified SB3 learner calls):
HipVecNormalize.step(actions) per control step, device -> host copies and info dicts included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drloco_amd.vec_env import vec_env
for n in (8, 256, 4096):
    env = vec_env(num_envs=n, seed=1)
    env.reset()
    rng = np.random.default_rng(0)
    acts = np.clip(0.5 * rng.standard_normal((50, n, 8)), -1, 1).astype(np.float32)
    for t in range(20):
        env.step(acts[t])
    t0 = time.perf_counter()
    K = 200
    for t in range(K):
        obs, rew, done, infos = env.step(acts[t % 50])
    dt = time.perf_counter() - t0
    print(f'{n:5d} envs: {dt / K * 1e6:8.1f} us per VecEnv.step  = {n * K / dt / 1e6:6.3f} M env-steps/s through the numpy API', flush=True)
    env.close()
