#!/usr/bin/env python3
"""Small driver for rocprofv3: N walkers, a few dozen control steps (keeps profiles short)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd.vec_env import HipVecEnv

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--steps', type=int, default=448)
ap.add_argument('--warm', type=int, default=64)
ap.add_argument('--single', action='store_true', help='one control step per launch (dl_step) instead of dl_rollout_fixed')
ap.add_argument('--variant', type=int, default=0, help='lanes per walker: 0 auto, 1, 16')
ap.add_argument('--walker', choices=['straight', 'loco3d'], default='straight')
ap.add_argument('--no-split', action='store_true', help='one wave per four walkers instead of the split workgroups bench.py uses for the straight walker')
args = ap.parse_args()
if args.walker == 'loco3d':
    from drloco_amd import mocap, models
    ang, vel = mocap.synthetic_loco3d(L=60000, seed=0)
    env = HipVecEnv(models.WALKER_165CM, num_envs=args.envs, lanes_per_walker=args.variant, refs=mocap.loco3d_table(ang, vel))
else:
    env = HipVecEnv(num_envs=args.envs, lanes_per_walker=args.variant)
if args.variant in (0, 16) and not args.no_split:
    env.set_split(True)          # the launch form of the benchmark (both walkers since round 5)
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(args.warm + args.steps, args.envs, env.nu, device='cuda', generator=g), -1, 1)
if args.single:
    for t in range(args.warm + args.steps):
        env.step_tensors(acts[t])
else:       # the benchmark's form: dl_rollout_fixed in the benchmark's launch schedule (one launch of T control steps)
    T = args.warm + args.steps
    env.rollout_fixed(acts[:T])             # ONE launch per rollout: bench.py's default schedule (T <= 512)
torch.cuda.synchronize()
print('done')
