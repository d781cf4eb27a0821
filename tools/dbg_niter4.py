import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from drloco_amd import lib
lib.LIB_PATH = os.path.join(ROOT, 'build_dbg', 'libdrloco_hip_dbg.so')
from drloco_amd.vec_env import HipVecEnv
st = dict(np.load(os.path.join(ROOT, 'build_dbg', 'state87.npz')))
env = HipVecEnv(num_envs=4096, reserved=1)
env.set_state(qpos=st['qpos'], qvel=st['qvel'], warm=st['warm'], cursor=st['cursor'], walked=st['walked'])
qa, nc, ne, ni = env.forward()
torch.cuda.synchronize()
print('niter', ni[836])
