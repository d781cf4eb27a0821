import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from drloco_amd import lib
from drloco_amd.vec_env import HipVecEnv
env = HipVecEnv(num_envs=4096, reserved=1)
env.reset_tensors()
g = torch.Generator(device='cuda'); g.manual_seed(4321)
acts = torch.clamp(0.5 * torch.randn(400, 4096, 8, device='cuda', generator=g), -1, 1)
for t in range(88):
    env.step_tensors(acts[t])
st = env.get_state()
np.savez(os.path.join(ROOT, 'build_dbg', 'state87.npz'), **st)
print('saved', env.forward()[3][836])
