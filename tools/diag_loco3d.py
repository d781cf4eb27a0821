#!/usr/bin/env python3
"""Throughput of the 19-dof walker (BASELINE config 4's walker) on the lane-per-walker kernels, synthetic loco3d table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd import mocap, models
from drloco_amd.vec_env import HipVecEnv
n, T = 4096, 64
ang, vel = mocap.synthetic_loco3d(L=60000, seed=0)
env = HipVecEnv(models.WALKER_165CM, num_envs=n, refs=mocap.loco3d_table(ang, vel))
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(1)
acts = torch.clamp(0.5 * torch.randn(T, n, 13, device='cuda', generator=g), -1, 1)
for t in range(8):
    env.step_tensors(acts[t])
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T):
    env.step_tensors(acts[t])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'walker165 (19 dof, frame_skip 10), {n} walkers, lane-per-walker kernels: {n * T / dt:.0f} env-steps/s, {dt / T * 1e3:.2f} ms per control step')
