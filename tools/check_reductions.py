#!/usr/bin/env python3
"""GAE / advantage statistics / VecNormalize reward branch against plain torch float64 restatements at the shapes training uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drloco_amd.rollout import HipRolloutBuffer
dev = torch.device('cuda')
def ref_gae(rew, val, start, last_val, last_done, gamma, lam):
    T, N = rew.shape
    adv = torch.zeros(T, N, dtype=torch.float64, device=dev); last = torch.zeros(N, dtype=torch.float64, device=dev)
    for t in reversed(range(T)):
        if t == T - 1: nnt = 1.0 - last_done.double(); nv = last_val.double()
        else: nnt = 1.0 - start[t + 1].double(); nv = val[t + 1].double()
        delta = rew[t].double() + gamma * nv * nnt - val[t].double()
        last = delta + gamma * lam * nnt * last
        adv[t] = last
    return adv, adv + val.double()
g = torch.Generator(device=dev); g.manual_seed(0)
for T, N in ((128, 128), (512, 4096), (100, 96), (4, 4096), (2048, 8), (129, 130)):
    buf = HipRolloutBuffer(T, N, 29, 8, dev)
    buf.rewards.copy_(torch.randn(T, N, device=dev, generator=g)); buf.values.copy_(torch.randn(T, N, device=dev, generator=g))
    buf.episode_starts.copy_((torch.rand(T, N, device=dev, generator=g) < 0.02).to(torch.uint8))
    lv = torch.randn(N, device=dev, generator=g); ld = (torch.rand(N, device=dev, generator=g) < 0.1).to(torch.uint8)
    adv, ret = buf.compute_returns_and_advantage(lv, ld)
    ra, rr = ref_gae(buf.rewards, buf.values, buf.episode_starts, lv, ld, 0.995, 0.95)
    e1 = float((adv.double() - ra).abs().max()); e2 = float((ret.double() - rr).abs().max())
    a0 = adv.clone()
    buf.normalize_advantages()
    ref_n = (a0.double() - a0.double().mean()) / (a0.double().std(unbiased=True) + 1e-8)
    e3 = float((buf.advantages.double() - ref_n).abs().max())
    print(f'T={T:5d} N={N:5d}: max |adv - ref| {e1:.2e}  |ret - ref| {e2:.2e}  |normalised adv - ref| {e3:.2e}  (scale {float(ra.abs().max()):.1f})', flush=True)

# ---- dl_vecnormalize_step against a float64 restatement of SB3 1.0 VecNormalize.step_wait, walker counts training uses
import ctypes as C
from drloco_amd import lib
from drloco_amd.vec_env import _ptr, _stream
L = lib.load()
for N in (128, 96, 8, 4096, 1000):
    D = 29
    mean = torch.zeros(D, dtype=torch.float64, device=dev); var = torch.ones(D, dtype=torch.float64, device=dev); cnt = torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
    rmean = torch.zeros(1, dtype=torch.float64, device=dev); rvar = torch.ones(1, dtype=torch.float64, device=dev); rcnt = torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
    ret = torch.zeros(N, dtype=torch.float64, device=dev)
    work = torch.zeros(abi.vn_workspace_bytes(D) // 8, dtype=torch.float64, device=dev)
    m, v, c = np.zeros(D), np.ones(D), 1e-4
    rm, rv, rc, rret = 0.0, 1.0, 1e-4, np.zeros(N)
    worst = 0.0
    for t in range(60):
        obs = (torch.randn(N, D, device=dev, generator=g) * 3 + 1).contiguous(); rew = torch.rand(N, device=dev, generator=g).contiguous()
        done = (torch.rand(N, device=dev, generator=g) < 0.05).to(torch.uint8)
        oo = torch.empty_like(obs); ro = torch.empty_like(rew)
        lib.check(L.dl_vecnormalize_step(_ptr(obs), _ptr(rew), _ptr(done), _ptr(mean), _ptr(var), _ptr(cnt), _ptr(ret), _ptr(rmean), _ptr(rvar), _ptr(rcnt),
                                         N, D, 0.99, 1e-8, 10.0, 10.0, 15, _ptr(oo), _ptr(ro), _ptr(work), _stream()))
        x = obs.double().cpu().numpy(); r = rew.double().cpu().numpy(); d = done.cpu().numpy().astype(bool)
        bm, bv = x.mean(0), x.var(0); tot = c + N; delta = bm - m
        m2 = v * c + bv * N + delta ** 2 * c * N / tot; m = m + delta * N / tot; v = m2 / tot; c = tot
        xo = np.clip((x - m) / np.sqrt(v + 1e-8), -10, 10)
        rret = rret * 0.99 + r
        bm, bv = rret.mean(), rret.var(); tot = rc + N; delta = bm - rm
        m2 = rv * rc + bv * N + delta ** 2 * rc * N / tot; rm = rm + delta * N / tot; rv = m2 / tot; rc = tot
        rn = np.clip(r / np.sqrt(rv + 1e-8), -10, 10)
        rret[d] = 0
        worst = max(worst, np.abs(oo.cpu().numpy() - xo).max(), np.abs(ro.cpu().numpy() - rn).max(), abs(float(rvar) - rv) / rv, np.abs(ret.cpu().numpy() - rret).max())
    print(f'VecNormalize N={N:5d}: worst deviation over 60 steps {worst:.2e}   (ret var {float(rvar):.4f} vs {rv:.4f})', flush=True)
