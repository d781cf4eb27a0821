#!/usr/bin/env python3
"""Do env-step kernels of several handles on several HIP streams overlap?  (They do when the streams sit on different hardware
queues -- two default-priority torch streams may share one; and two handles of 4096 walkers on two streams reach the mean-wave
bound with single-step launches, because the second handle's workgroups fill the SIMDs the first one's fast waves leave.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drloco_amd.vec_env import HipVecEnv
dev = torch.device('cuda', 0)
def run(G, B, K=64, prio=None):
    envs = [HipVecEnv(num_envs=B, seed=1234, env_index_base=g * B) for g in range(G)]
    streams = [torch.cuda.Stream(device=dev, priority=(prio[g] if prio else 0)) for g in range(G)]
    acts = [torch.clamp(0.5 * torch.randn(K, B, 8, device=dev), -1, 1) for _ in range(G)]
    for e in envs: e.reset_tensors()
    torch.cuda.synchronize()
    def go():
        for t in range(K):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    envs[g].step_tensors(acts[g][t])
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{G} handle(s) x {B} walkers on {G} stream(s){" prio " + str(prio) if prio else ""}: {dt / K * 1e6:.1f} us per control step of all handles, {G * B * K / dt / 1e6:.2f} M env-steps/s', flush=True)
    for e in envs: e.close()
run(1, 4096); run(1, 2048); run(2, 2048); run(2, 2048, prio=[0, -1]); run(4, 1024); run(2, 4096)
