import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.vec_env import HipVecEnv
for variant in (1, 0):
    env = HipVecEnv(num_envs=4096, reserved=variant)
    env.reset_tensors()
    g = torch.Generator(device='cuda'); g.manual_seed(4321)
    acts = torch.clamp(0.5 * torch.randn(200, 4096, 8, device='cuda', generator=g), -1, 1)
    hist = np.zeros(128, int); mx = []; times = []
    for t in range(200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.step_tensors(acts[t])
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        if t % 4 == 0:
            _, nc, ne, ni = env.forward()
            hist += np.bincount(ni, minlength=128)[:128]; mx.append(int(ni.max()))
    times = np.array(times[20:]) * 1e3
    print('variant', variant, 'niter hist', hist[:16], 'tail>=16', hist[16:].sum(), 'max per sample', sorted(mx)[-8:])
    print('  step ms: mean', times.mean(), 'median', np.median(times), 'p90', np.quantile(times, .9), 'max', times.max())
