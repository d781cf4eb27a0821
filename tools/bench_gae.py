#!/usr/bin/env python3
"""Time dl_gae on the benchmark shape [512, 4096]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd.rollout import HipRolloutBuffer
T, N = 512, 4096
buf = HipRolloutBuffer(T, N, 29, 8, 'cuda')
buf.rewards.uniform_(0, 1.2); buf.values.normal_(); lv = torch.randn(N, device='cuda'); ld = torch.zeros(N, dtype=torch.uint8, device='cuda')
for _ in range(5): buf.compute_returns_and_advantage(lv, ld)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): buf.compute_returns_and_advantage(lv, ld)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f'dl_gae [{T}, {N}]: {us:.1f} us, {20 * T * N / us / 1e6:.2f} TB/s of 20 B per sample')
