#!/usr/bin/env python3
"""What the hot path does while the walkers WALK: the packaged trained policy (drloco_amd/data/walking_policy.npz: 3000-step episodes, ~22 m per
episode) against the random-init policy of `bench.py --policy`, both through the same rollouts (4096 walkers x 512 steps, dl_collect_rollouts, sampled actions as in
training).  Per policy: time per control step of each rollout form, Newton iterations and constraint rows per forward evaluation (mean over the timed rollouts, 99 %
quantile over walkers, and over the single evaluations of the rollouts' last control steps), histogram of active contacts (dl_forward at the state after every rollout),
episode statistics of the Monitor.  With DL_LIB_PATH = a -DDL_EXP_ROLLOUT_PROF=2 build the exact mode's phases per workgroup and control step as well (slowest / mean
env phase: what the per-step coupling costs).  usage: python3 tools/diag_walking.py [rollouts]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from drloco_amd import checkpoint, lib
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize

n, T = 4096, 512
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
WARM = 8          # rollouts before anything is measured: 4096 steps, every walker past its first episode boundary or at the limit
prof_build = 'prof' in os.environ.get('DL_LIB_PATH', '')


def run(label, trained, form, moments):
    venv = HipVecEnv(num_envs=n, seed=1234)
    venv.set_split(True)
    vn = HipVecNormalize(venv); vn.reset()
    if trained:
        pol, _ = checkpoint.load_walking_policy(vec_normalize=vn, seed=99)
        vn.norm_obs_t.copy_(venv.obs); vn._normalize_obs_inplace(vn.norm_obs_t)
        restore = checkpoint.moment_seat(vn)          # every rollout from the checkpoint's moments: a fixed policy under free-running statistics drifts out of its input distribution
    else:
        restore = lambda: None
        pol = HipPolicy(hidden=512, seed=99)
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
    last_obs, last_done = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
    persistent = form == 'persistent'
    for _ in range(WARM):
        restore()
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=persistent, moments=moments)
    venv.debug_counters()
    ncon_hist, rows_hist, it_hist = np.zeros(19, np.int64), np.zeros(130, np.int64), np.zeros(128, np.int64)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms, ends = [], 0
    prof = torch.zeros((n // 16) * 4 * 11 + 512 * (n // 16) * 4, dtype=torch.int64, device='cuda') if prof_build else None
    tails = []
    for r in range(R):
        restore()
        ev0.record()
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=persistent, moments=moments)
        ev1.record(); torch.cuda.synchronize()
        ms.append(ev0.elapsed_time(ev1))
        ends += int(buf._starts[1:T + 1].sum().item())
        it_last, rows_last = venv.debug_eval_iters(rows=True)
        it_hist += np.bincount(it_last.reshape(-1), minlength=128)[:128]; rows_hist += np.bincount(np.clip(rows_last.reshape(-1), 0, 129), minlength=130)
        _, ncon, _, _ = venv.forward()
        ncon_hist += np.bincount(np.clip(ncon, 0, 18), minlength=19)
        if prof is not None and persistent and moments == 'per_step':
            lib.check(venv._lib.dl_debug_rollout_prof(venv._h, C.c_void_p(prof.data_ptr()), None))
            ps = prof[(n // 16) * 4 * 11:].view(512, n // 16, 4)[:T].cpu().numpy().astype(np.float64)
            E = ps[..., 1]
            if (ps[..., 0] > 0).all():
                tails.append((E.mean() / 100, E.max(1).mean() / 100, np.quantile(E.max(1) / E.mean(1), 0.99)))
    cnt = venv.debug_counters(clear=False).astype(np.float64)
    evals = T * R * 4 * venv.model.frame_skip
    q = lambda h, p: int(np.searchsorted(np.cumsum(h) / h.sum(), p))
    ep_len = float(np.mean(venv.get_attr('ep_len_smoothed'))); moved = float(np.mean(venv.get_attr('moved_distance'))); mrew = float(np.mean(venv.get_attr('mean_reward_smoothed')))
    out = dict(label=label, form=form + ('' if moments == 'per_step' else ' / per-rollout moments'), ms=float(np.mean(ms)), msteps=n * T / np.mean(ms) / 1e3,
               it=cnt[0].mean() / evals, it_p99w=float(np.quantile(cnt[0] / evals, 0.99)), rows=cnt[2].mean() / evals, rows_p99w=float(np.quantile(cnt[2] / evals, 0.99)),
               it_p50=q(it_hist, 0.5), it_p99=q(it_hist, 0.99), it_max=int(np.nonzero(it_hist)[0].max()), rows_p50=q(rows_hist, 0.5), rows_p99=q(rows_hist, 0.99), rows_max=int(np.nonzero(rows_hist)[0].max()),
               ncon=ncon_hist / ncon_hist.sum(), ncon_mean=float((ncon_hist * np.arange(19)).sum() / ncon_hist.sum()), ends=ends / R, diverged=int(cnt[3].sum()),
               ep_len=ep_len, moved=moved, mrew=mrew, tails=np.array(tails).mean(0) if tails else None)
    venv.close()
    return out


rows = []
run('warm-up handle (discarded)', False, 'persistent', 'per_step') if R > 1 else None
for trained in (False, True):
    label = 'trained policy (walking)' if trained else 'random-init policy'
    for form, moments in (('persistent', 'per_step'), ('persistent', 'per_rollout'), ('launches', 'per_step')):
        rows.append(run(label, trained, form, moments))
print(f'{n} walkers x {T}-step rollouts, {R} timed rollouts after {WARM} warm-up rollouts, split workgroups, sampled actions; library {os.environ.get("DL_LIB_PATH", "product")}')
print(f'{"policy":26s} {"rollout form":36s} {"ms/rollout":>10s} {"M steps/s":>10s} {"us/ctrl step":>12s} {"iter/eval":>9s} {"p99 walker":>10s} {"rows/eval":>9s} {"p99 walker":>10s} {"ep ends/rollout":>15s} {"diverged":>8s}')
for r in rows:
    print(f'{r["label"]:26s} {r["form"]:36s} {r["ms"]:10.2f} {r["msteps"]:10.2f} {r["ms"] / T * 1e3:12.1f} {r["it"]:9.3f} {r["it_p99w"]:10.3f} {r["rows"]:9.2f} {r["rows_p99w"]:10.2f} {r["ends"]:15.1f} {r["diverged"]:8d}')
for r in rows[0::3]:
    print(f'\n{r["label"]}: Monitor: smoothed episode length {r["ep_len"]:.0f} control steps, walked {r["moved"]:.2f} m per episode, mean step reward {r["mrew"]:.3f}')
    print(f'  single evaluations (all 20 of the last control step of every timed rollout): Newton iterations p50 {r["it_p50"]} / p99 {r["it_p99"]} / max {r["it_max"]};  constraint rows p50 {r["rows_p50"]} / p99 {r["rows_p99"]} / max {r["rows_max"]}')
    print(f'  active contacts per evaluation (dl_forward at the state after each rollout): mean {r["ncon_mean"]:.2f};  ' + '  '.join(f'{c}: {f:.3f}' for c, f in enumerate(r['ncon']) if f >= 0.0005))
    if r['tails'] is not None:
        print(f'  exact mode, env phase per workgroup and control step [us]: mean {r["tails"][0]:.1f}, slowest workgroup {r["tails"][1]:.1f} = {r["tails"][1] / r["tails"][0]:.3f} x mean (p99 over steps of slowest / mean: {r["tails"][2]:.2f})')
a, b = rows[0], rows[3]
print(f'\nwalking / random-init (persistent, exact): time {b["ms"] / a["ms"]:.3f}, iterations per evaluation {b["it"] / a["it"]:.3f}, rows per evaluation {b["rows"] / a["rows"]:.3f}, contacts {b["ncon_mean"] / max(a["ncon_mean"], 1e-9):.2f}')
