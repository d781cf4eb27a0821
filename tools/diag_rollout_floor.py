#!/usr/bin/env python3
"""How far is the exact mode of the persistent rollout kernel (SB3's per-step VecNormalize: every workgroup meets every other workgroup at every control
step) from its STRUCTURAL floor?  Needs a -DDL_EXP_ROLLOUT_PROF=2 build (DL_LIB_PATH): the kernel then records, per control step and workgroup, the time (s_memrealtime: the constant
100 MHz counter -- the XCDs' shader clocks differ, and these records are compared across workgroups) of the policy phase P, the env phase E and the moment sums +
exchange R (waiting included).
  floor      = sum_t [ min_wg P_t + max_wg E_t + min_wg R_t ]     -- with per-step coupling a step cannot end before its slowest workgroup's env phase; the
               policy phase and the exchange are the same work for every workgroup (their minimum over workgroups = the work without waiting)
  measured   = max_wg sum_t (P + E + R + waiting in the exchange)  -- what the launch takes
  free-run   = max_wg sum_t (min P_t + E_t[wg])                   -- what the per-rollout relaxation's structure would take with these env phases
The benchmark's rollout: 4096 walkers x 512 steps, bench.py's policy seed and action noise.  usage: DL_LIB_PATH=build_variants/libdrloco_hip_prof.so python3 tools/diag_rollout_floor.py"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from drloco_amd import lib
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

n, T, R = 4096, 512, 3
nblk = n // 16
venv = HipVecEnv(num_envs=n, seed=1234)
vn = HipVecNormalize(venv); vn.reset()
pol = HipPolicy(hidden=512, seed=99)
buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
last_obs, last_done = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
prof = torch.zeros(nblk * 4 * 11 + 512 * nblk * 4, dtype=torch.int64, device='cuda')
rows = []
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(R + 1):
    ev0.record()
    buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
    ev1.record()
    torch.cuda.synchronize()
    lib.check(venv._lib.dl_debug_rollout_prof(venv._h, C.c_void_p(prof.data_ptr()), None))
    if r == 0:
        continue          # warm-up rollout (all walkers start together)
    ps = prof[nblk * 4 * 11:].view(512, nblk, 4)[:T].cpu().numpy().astype(np.float64)          # [T, wg, (P, E, R, wait in R)]
    P, E, Rr, Wt = ps[..., 0], ps[..., 1], ps[..., 2], ps[..., 3]          # slot 2: block sums + delivering them (and the merge after the exchange), slot 3: waiting for the grid
    if not (P > 0).all():
        sys.exit('no per-step records: is DL_LIB_PATH a -DDL_EXP_ROLLOUT_PROF build?')
    floor = (P.min(1) + E.max(1) + Rr.min(1)).sum()
    measured = (P + E + Rr + Wt).sum(0).max()
    free = (P.min(1)[:, None] + E).sum(0).max()
    ms = ev0.elapsed_time(ev1)
    rows.append((measured, floor, free, ms, E.mean(), E.max(1).mean(), P.min(1).mean(), Rr.min(1).mean(), Wt.mean()))
    print(f'rollout {r}: measured {measured / 1e5:8.2f} ms ({ms:6.2f} ms by events incl. launch)   floor {floor / 1e5:8.2f} ms   measured / floor {measured / floor:.3f}   '
          f'free-running structure {free / 1e5:8.2f} ms ({free / measured:.3f} x measured)')
    print(f'           per step: [us] env phase mean over workgroups {E.mean() / 100:8.2f}, slowest workgroup {E.max(1).mean() / 100:8.2f} ({E.max(1).mean() / E.mean():.3f} x mean); policy phase {P.min(1).mean() / 100:7.2f}; sums + exchange without waiting {Rr.min(1).mean() / 100:6.2f}; mean wait in the exchange {Wt.mean() / 100:7.2f}')
a = np.array(rows).mean(0)
print(f'mean of {R} rollouts: measured / floor = {a[0] / a[1]:.3f}   (floor {a[1] / a[0] * a[3]:.1f} ms of {a[3]:.1f} ms)   slowest / mean env phase {a[5] / a[4]:.3f}')
venv.close()
