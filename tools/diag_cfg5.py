#!/usr/bin/env python3
"""Diagnose the float32 outliers of test_f32_randomization_and_push_schedule_vs_oracle: per control step, the walkers beyond the one-step
tolerance, their randomisation, per-evaluation iteration counts (float32 vs float64 build) and forward-dynamics agreement at the step's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from drloco_amd import abi, mocap, models
from drloco_amd.vec_env import HipVecEnv
lanes = sys.argv[1] if len(sys.argv) > 1 else 'split'
lanes = 16 if lanes == '16' else lanes
model, refs = models.make_model(), mocap.RefTable.load()
n, K, period, dur = 2048, 8, 4, 2
rng = np.random.default_rng(21)
ms = rng.uniform(0.8, 1.2, n).astype(np.float32); fr = rng.uniform(0.5, 1.1, n).astype(np.float32)
ang = rng.uniform(0, 2 * np.pi, n)
force = np.stack([50 * np.cos(ang), 50 * np.sin(ang), np.zeros(n)], 1).astype(np.float32); force[::5] = 0
phase = rng.integers(0, period, n).astype(np.int32)
dev = HipVecEnv(num_envs=n, precision=32, model=model, refs=refs, lanes_per_walker=lanes)
d64 = HipVecEnv(num_envs=n, precision=64, model=model, refs=refs, lanes_per_walker=16)
orc = O.OracleEnv(model, refs, dev.cfg, n)
steps = rng.integers(0, 30, n).astype(np.int32)
pos = (rng.random(n) * refs.step_len[steps]).astype(np.int32)
orc.reset(init_step=steps, init_pos=pos)
orc.set_randomization(ms.astype(np.float64), fr.astype(np.float64))
for e in (dev, d64):
    e.reset(init_step=steps, init_pos=pos); e.set_randomization(ms, fr); e.debug_counters()
for t in range(12):
    orc.step(np.clip(0.3 * rng.standard_normal((n, 8)), -1, 1))
for e in (dev, d64):
    e.set_push_schedule(force, phase, period, dur)
def sync(e):
    st = orc.get_state(); e.set_state(qpos=st['qpos'], qvel=st['qvel'], warm=st['warm'], cursor=st['cursor'], walked=st['walked']); return st
for k in range(K):
    st0 = sync(dev); sync(d64)
    on = ((k + phase) % period) < dur
    orc.set_randomization(xfrc=(force * on[:, None]).astype(np.float64))
    a = np.clip(0.5 * rng.standard_normal((n, 8)), -1, 1).astype(np.float32)
    o1, r1, d1, term1, _ = orc.step(a.astype(np.float64)); o2, r2, d2, _ = dev.step(a); o3, r3, d3, _ = d64.step(a)
    c32, c64 = dev.debug_counters(), d64.debug_counters()
    it32, it64 = dev.debug_eval_iters(), d64.debug_eval_iters()
    s1, s2 = orc.get_state(), dev.get_state()
    live = ~d2
    dq = np.abs(s1['qpos'] - s2['qpos']).max(0); dv = (np.abs(s1['qvel'] - s2['qvel']) / (1 + np.abs(s1['qvel']))).max(0)
    rel = np.abs(r1 - r2) / np.maximum(np.abs(r1), 1e-9)
    bad = live & ((dq > 2e-4) | (dv > 5e-3) | (rel > 1e-4))
    print(f'step {k}: {int(bad.sum())} walkers beyond the one-step tolerance; rows differ on {int((c32[2] != c64[2]).sum())}')
    for w in np.nonzero(bad)[0]:
        print(f'   walker {w}: mscale {ms[w]:.3f} mu {fr[w]:.3f} pushed {bool(on[w] and np.abs(force[w]).sum() > 0)}  dq {dq[w]:.2e} dv {dv[w]:.2e} rel {rel[w]:.2e}  rows f32 {c32[2][w]} f64 {c64[2][w]}  iters f32 {it32[:, w].tolist()} f64 {it64[:, w].tolist()}')
        print(f'      start: qpos[2] {st0["qpos"][2, w]:.4f} |qvel| max {np.abs(st0["qvel"][:, w]).max():.2f}  trunk angles {st0["qpos"][3:6, w].round(3).tolist()}  end (oracle) qvel max {np.abs(s1["qvel"][:, w]).max():.2f}; worst dof dv {int(np.argmax(np.abs(s1["qvel"][:, w] - s2["qvel"][:, w])))}')
