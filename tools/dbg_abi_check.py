"""Bit-exact comparison of the production build (forward evaluation behind a call) with a build
whose forward evaluation is inlined (-DDL_INLINE_FORWARD): same arithmetic, different register
allocation / ABI.  Any difference points at a compiler or ABI problem."""
import os, sys, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

def run(libpath, prec):
    code = f'''
import sys; sys.path.insert(0, {ROOT!r})
import numpy as np, torch
from drloco_amd import lib
lib.LIB_PATH = {libpath!r}
from drloco_amd.vec_env import HipVecEnv
env = HipVecEnv(num_envs=1024, precision={prec})
env.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(7)
acts = torch.clamp(0.5*torch.randn(120, 1024, 8, device="cuda", generator=g), -1, 1)
obs, rew, done = env.rollout_fixed(acts)
st = env.get_state()
np.savez({libpath!r} + ".{prec}.npz", obs=obs.cpu().numpy(), rew=rew.cpu().numpy(), done=done.cpu().numpy(), **st,
         stats=np.array(env.get_attr("ep_ret_smoothed")))
'''
    subprocess.check_call([sys.executable, '-c', code])
    return dict(np.load(libpath + f'.{prec}.npz'))

a_lib = os.path.join(ROOT, 'drloco_amd', 'csrc', 'libdrloco_hip.so')
b_lib = os.path.join(ROOT, 'build_dbg', 'libdrloco_hip_inline.so')
for prec in (32, 64):
    A, B = run(a_lib, prec), run(b_lib, prec)
    for k in A:
        same = np.array_equal(A[k], B[k], equal_nan=True)
        print(prec, k, 'identical' if same else f'DIFFERENT max abs diff {np.nanmax(np.abs(A[k].astype(np.float64) - B[k].astype(np.float64)))}')
