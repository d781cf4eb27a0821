#!/usr/bin/env python3
"""Time dl_policy_forward (29-512-512-{8,1}) for several batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd.policy import HipPolicy
pol = HipPolicy()
import ctypes as C
from drloco_amd import lib
from drloco_amd.vec_env import _ptr, _stream
for n in (16, 1024, 4096, 16384):
  for mode in (0, 1):          # 0: torch's weight layout, 1: the packed copy
    obs = torch.randn(n, 29, device='cuda')
    a = torch.empty(n, 8, device='cuda'); v = torch.empty(n, device='cuda'); lp = torch.empty(n, device='cuda'); p = pol._params()
    pk = pol._packed_weights() if mode else None
    call = lambda: lib.check(pol._lib.dl_policy_forward_packed(C.byref(p), _ptr(pk), _ptr(obs), n, None, 1, 1, 0, 0, _ptr(a), _ptr(v), _ptr(lp), _stream()))
    for _ in range(20): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f'mode {mode} policy forward {n:6d} rows: {us:7.1f} us per call, {2 * n * (29 * 512 + 512 * 512 + 512 * 9) / us / 1e6:6.1f} TFLOP/s')
