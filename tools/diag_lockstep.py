#!/usr/bin/env python3
"""What does the lockstep of the four walkers of a wave cost, and what would re-grouping recover?  (offline experiment, no kernel change.)

The benchmark rollout (4096 walkers x 512 control steps, bench.py's action tape, split workgroups) is taken one control step per launch with
the diagnostics on; after every step the Newton iterations of every walker in each of the step's 20 forward evaluations are read back
(dl_debug_eval_iters).  A wave runs max-over-its-walkers iterations per evaluation, so per control step its solver work is
    sum over the 20 evaluations of max over the wave's 4 walkers of iters(walker, evaluation).
That sum is evaluated for
  (i)    today's static grouping (walkers 4k .. 4k + 3),
  (ii)   the walkers of a 16-walker workgroup re-grouped before every control step by the PREVIOUS step's iteration sum (sorted, 4 per wave)
         -- what the kernel could do: the 16 walkers of a split workgroup share one LDS,
  (ii')  the same with other predictors (previous step's max, an exponential average, "was in the air last step"),
  (iii)  the oracle grouping: sorted by THIS step's sum (the bound of any per-step re-grouping by a scalar), and
  (iv)   no lockstep at all (every walker pays its own iterations): the floor.
usage (GPU box): python3 tools/diag_lockstep.py [--steps 512] [--save gpurun_out/lockstep.npz]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def wave_cost(it, order=None, block=16):
    """it: [E, N] iterations per evaluation; order: [N] permutation inside blocks of `block` walkers (None: identity) -> sum over waves and evaluations of the max over a wave's 4 walkers"""
    x = it if order is None else it[:, order]
    E, N = x.shape
    return int(x.reshape(E, N // 4, 4).max(axis=2).sum())


def block_sort(key, block=16):
    """permutation that sorts the walkers of every block of `block` by key (stable)"""
    N = key.shape[0]
    k = key.reshape(N // block, block)
    o = np.argsort(k, axis=1, kind='stable') + (np.arange(N // block) * block)[:, None]
    return o.reshape(-1)


def analyse(its, blocks=(16, 32, 64)):
    """its: [T, E, N] int.  Prints the table; returns a dict of totals."""
    T, E, N = its.shape
    tot = {}
    tot['floor (no lockstep)'] = its.sum() / 4.0
    tot['static'] = sum(wave_cost(its[t]) for t in range(T))
    for B in blocks:
        prev_sum = np.zeros(N); prev_max = np.zeros(N); ema = np.zeros(N)
        c_prev = c_pmax = c_ema = c_orc = c_orc_max = 0
        for t in range(T):
            s, mx = its[t].sum(0), its[t].max(0)
            c_prev += wave_cost(its[t], block_sort(prev_sum, B))
            c_pmax += wave_cost(its[t], block_sort(prev_max * 1000 + prev_sum, B))
            c_ema += wave_cost(its[t], block_sort(ema, B))
            c_orc += wave_cost(its[t], block_sort(s, B))
            c_orc_max += wave_cost(its[t], block_sort(mx * 1000 + s, B))
            prev_sum, prev_max = s, mx
            ema = 0.5 * ema + 0.5 * s
        tot[f'regroup in blocks of {B}: by previous step sum'] = c_prev
        tot[f'regroup in blocks of {B}: by previous step max, then sum'] = c_pmax
        tot[f'regroup in blocks of {B}: by exponential average of the sums'] = c_ema
        tot[f'regroup in blocks of {B}: ORACLE (this step sum)'] = c_orc
        tot[f'regroup in blocks of {B}: ORACLE (this step max, then sum)'] = c_orc_max
    base = tot['static']
    print(f'{T} control steps x {E} evaluations x {N} walkers; mean iterations per walker and evaluation {its.mean():.3f}; wave iterations per evaluation, static grouping {base / (T * E * N / 4):.3f}')
    for k, v in tot.items():
        print(f'  {k:66s} {v / (T * E * N / 4):7.3f} wave iterations per evaluation   {100 * (v - base) / base:+6.1f} % vs static')
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--save', type=str, default='')
    ap.add_argument('--load', type=str, default='', help='analyse a saved dump instead of running (no GPU needed)')
    args = ap.parse_args()
    if args.load:
        its = np.load(args.load)['iters'].astype(np.int32)
        analyse(its)
        return
    import torch
    from bench import tape_normal
    from drloco_amd.vec_env import HipVecEnv
    n, T = args.envs, args.steps
    env = HipVecEnv(num_envs=n, seed=1234, lanes_per_walker='split')
    acts = torch.clamp(0.5 * tape_normal(4321, 0, T, 0, n, env.nu, env.device), -1, 1)
    env.reset_tensors()
    env.debug_counters()
    its = np.zeros((T, 4 * env.model.frame_skip, n), np.int8)
    for t in range(T):
        env.step_tensors(acts[t])
        its[t] = env.debug_eval_iters()
    if args.save:
        np.savez_compressed(args.save, iters=its)
    analyse(its.astype(np.int32))
    # where the iterations are: distribution of the per-evaluation counts
    h = np.bincount(its.reshape(-1).astype(np.int64), minlength=12)
    print('iterations per evaluation, histogram 0..11+:', (h[:11] / h.sum()).round(4).tolist(), float(h[11:].sum() / h.sum()))
    s = its.astype(np.int32).sum(1)        # [T, N] per control step
    print('per control step and walker: mean %.2f, median %d, q90 %d, q99 %d, max %d' % (s.mean(), np.median(s), np.quantile(s, 0.9), np.quantile(s, 0.99), s.max()))
    # persistence: correlation of a walker's step sum with its previous step's
    a, b = s[1:].reshape(-1).astype(np.float64), s[:-1].reshape(-1).astype(np.float64)
    print('correlation of a walker\'s iteration sum with its previous step\'s: %.3f' % np.corrcoef(a, b)[0, 1])
    env.close()


if __name__ == '__main__':
    main()
