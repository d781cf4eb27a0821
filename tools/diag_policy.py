#!/usr/bin/env python3
"""Where the forward pass of the policy goes (needs a -DDL_EXP_POL_PROF build, selected with DL_LIB_PATH): shader-clock stamps of workgroup 0's
first and last wave at the section boundaries, stand-alone launch of 4096 rows (one workgroup per CU, the persistent kernel's situation)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd import lib
from drloco_amd.policy import HipPolicy
from drloco_amd.vec_env import _ptr, _stream
pol = HipPolicy()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
obs = torch.randn(n, 29, device='cuda')
a = torch.empty(n, 8, device='cuda'); v = torch.empty(n, device='cuda'); lp = torch.empty(n, device='cuda'); p = pol._params()
pk = pol._packed_weights()
L = pol._lib
acc = np.zeros((2, 8))
R = 50
for i in range(R + 5):
    lib.check(L.dl_policy_forward_packed(C.byref(p), _ptr(pk), _ptr(obs), n, None, 1, 1, 0, 0, _ptr(a), _ptr(v), _ptr(lp), _stream()))
    torch.cuda.synchronize()
    out = (C.c_longlong * 16)()
    assert L.dl_debug_pol_prof(out) == 0
    s = np.array(list(out), dtype=np.float64).reshape(2, 8)
    if i >= 5: acc += s - s[0, 0]
acc /= R
names = ['start', 'obs staged + barrier', 'layer 1 done, h1 staged', 'after the h1 barrier', 'hidden layer done', 'heads done', 'after the heads barrier', 'end']
for k in range(8):
    print(f'{names[k]:28s} wave 0: {acc[0, k]:8.0f}   wave 7: {acc[1, k]:8.0f}')
