#!/usr/bin/env python3
"""Where do the waves of one env-step launch land?  Needs an experiment build whose step kernel writes HW_ID / XCC_ID into the debug counters
(rows 1 and 3) -- selected with DL_LIB_PATH.  Prints how many waves share a SIMD and how many CUs / SIMDs are used.
The experiment build: in dl_group_env.hpp, where the step kernel updates st.dbg, store
    st.dbg[n + w] = (int)__builtin_amdgcn_s_getreg(0xF804);              // HW_ID, all 32 bits
    st.dbg[3 * n + w] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
instead of the maximum iteration count / the diverged-step count, add __attribute__((amdgpu_waves_per_eu(2, 2))) to k_env_step_g16 for
the register-capped variant, and build with the product flags into build_variants/."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.vec_env import HipVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = HipVecEnv(num_envs=n, seed=1)
env.reset_tensors(); env.debug_counters()
acts = torch.zeros(1, n, env.nu, device='cuda')
env.rollout_fixed(acts)
c = env.debug_counters()
hw, xcc = c[1][::4].astype(np.int64), c[3][::4].astype(np.int64)          # one entry per wave (4 walkers)
wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
xid = xcc & 15
per_simd = collections.Counter(zip(xid, se, sh, cu, simd))
per_cu = collections.Counter(zip(xid, se, sh, cu))
print(f'{len(hw)} waves on {len(per_cu)} CUs / {len(per_simd)} SIMDs; waves per SIMD: {dict(collections.Counter(per_simd.values()))}; waves per CU: {dict(sorted(collections.Counter(per_cu.values()).items()))}')
print('hw_id sample', [hex(int(x)) for x in hw[:6]], 'xcc', xid[:6])
