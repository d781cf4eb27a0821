import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from drloco_amd import models, lib
from drloco_amd.vec_env import HipVecEnv
for iters in (100, 6, 4, 2, 1, 0):
    m = models.make_model(); m.iterations = iters
    env = HipVecEnv(num_envs=4096, model=m)
    env.reset_tensors()
    g = torch.Generator(device='cuda'); g.manual_seed(4321)
    acts = torch.clamp(0.5 * torch.randn(160, 4096, 8, device='cuda', generator=g), -1, 1)
    for t in range(60): env.step_tensors(acts[t])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(60, 160): env.step_tensors(acts[t])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print('iterations', iters, 'ms/step', dt * 1e3, 'mean rew', env.rew.mean().item(), 'done frac', env.done.float().mean().item())
