"""World-size-2 gloo tests (CPU) of the N > 1 path, driving the PRODUCT's collective code (drloco_amd/collectives.py,
merge_moments_across_ranks) -- the same functions HipRolloutBuffer / HipVecNormalize / examples/train_ppo.py call over RCCL.

The rollout shards by walker index with no data-path collective.  On CPU the decomposition is exercised with the oracle
standing in for the step kernels and torch for the two reduction kernels: (1) the RSI stream is keyed by the GLOBAL walker
index, so two shards of 8 walkers reproduce one 16-walker run bit for bit; (2) C1: collectives.normalize_advantages over
the shards gives the normalisation of the concatenated batch; (3) C3: the exact moment merge; (4) C2: two ranks x N/2
walkers take the same PPO optimiser steps as one rank x N (flat-bucket gradient all-reduce, global minibatch statistics)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:          # (a pid-derived port can still be held by an earlier run's lingering socket)
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from drloco_amd import abi, mocap, models
    from oracle import oracle as O
    model, refs = models.make_model(), mocap.RefTable.load()
    n_shard, T = 8, 12
    env = O.OracleEnv(model, refs, abi.default_config(seed=77, env_index_base=rank * n_shard), n_shard)
    obs0 = env.reset()
    rng = np.random.default_rng(5)                     # same stream on every rank; each takes its slice
    adv_chunks, obs_last = [], None
    for t in range(T):
        a = np.clip(0.5 * rng.standard_normal((world * n_shard, 8)), -1, 1)
        obs_last, rew, done, _, _ = env.step(a[rank * n_shard:(rank + 1) * n_shard])
        adv_chunks.append(rew)
    adv = np.stack(adv_chunks).astype(np.float32)
    # C1 through the product's function (HipRolloutBuffer.normalize_advantages calls the same one with the two device kernels
    # in place of the torch stand-ins)
    from drloco_amd import collectives
    tn = torch.as_tensor(adv).double()
    collectives.normalize_advantages(tn, collectives.torch_adv_stats, collectives.torch_adv_apply)
    norm = tn.numpy()
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), obs0=obs0, obs_last=obs_last, adv=adv, norm=norm,
             cursor=env.get_state()['cursor'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_rollout_and_advnorm_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from drloco_amd import abi, mocap, models
    from oracle import oracle as O
    model, refs = models.make_model(), mocap.RefTable.load()
    n_shard, T = 8, 12
    env = O.OracleEnv(model, refs, abi.default_config(seed=77, env_index_base=0), world * n_shard)
    obs0 = env.reset()
    rng = np.random.default_rng(5)
    advs = []
    for t in range(T):
        a = np.clip(0.5 * rng.standard_normal((world * n_shard, 8)), -1, 1)
        obs_last, rew, done, _, _ = env.step(a)
        advs.append(rew)
    adv = np.stack(advs).astype(np.float32)
    want_norm = (adv.astype(np.float64) - adv.mean(dtype=np.float64)) / (adv.astype(np.float64).std(ddof=1) + 1e-8)
    cur = env.get_state()['cursor']
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        sl = slice(r * n_shard, (r + 1) * n_shard)
        assert np.array_equal(z['obs0'], obs0[sl])                 # same RSI draws as the unsharded run
        assert np.array_equal(z['obs_last'], obs_last[sl])
        assert np.array_equal(z['cursor'], cur[:, sl])
        assert np.array_equal(z['adv'], adv[:, sl])
        np.testing.assert_allclose(z['norm'], want_norm[:, sl], rtol=1e-9, atol=1e-9)


def _moments_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from drloco_amd.vec_env import merge_moments_across_ranks
    from oracle import oracle as O
    D = 5
    rng = np.random.default_rng(100)                   # the same data on every rank; each consumes its slice
    mean = np.zeros(D); var = np.ones(D); cnt = 1e-4
    tm = torch.zeros(D, dtype=torch.float64); tv = torch.ones(D, dtype=torch.float64); tc = torch.full((1,), 1e-4, dtype=torch.float64)
    sync = (tm.clone(), tv.clone(), tc.clone())
    for rollout in range(3):
        for t in range(4):
            x = rng.standard_normal((world, 16, D)) * np.arange(1, D + 1) + rollout
            cnt = O.moments_update(mean, var, cnt, x[rank])
        tm.copy_(torch.as_tensor(mean)); tv.copy_(torch.as_tensor(var)); tc.fill_(cnt)
        merge_moments_across_ranks(tm, tv, tc, *sync)
        mean[:] = tm.numpy(); var[:] = tv.numpy(); cnt = float(tc.item())
    np.savez(os.path.join(tmp, f'mom{rank}.npz'), mean=mean, var=var, cnt=cnt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_moment_merge_across_ranks(tmp_path):
    """C3: after a merge every rank holds the moments one process would have computed from all ranks' batches."""
    world, port = 2, _free_port()
    mp.spawn(_moments_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    D = 5
    rng = np.random.default_rng(100)
    xs = []
    for rollout in range(3):
        for t in range(4):
            xs.append((rng.standard_normal((world, 16, D)) * np.arange(1, D + 1) + rollout).reshape(-1, D))
    allx = np.concatenate(xs)
    n = 1e-4 + len(allx)
    # the prior (mean 0, var 1, count 1e-4) enters once
    want_mean = allx.sum(0) / n
    want_var = ((allx ** 2).sum(0) + 1e-4 * 1.0) / n - want_mean ** 2
    for r in range(world):
        z = np.load(tmp_path / f'mom{r}.npz')
        np.testing.assert_allclose(z['mean'], want_mean, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(z['var'], want_var, rtol=1e-9)
        assert abs(float(z['cnt']) - n) < 1e-9


# ---- C2: the gradient step --------------------------------------------------------------------------------------
def _ppo_problem(n_global=16, T=8, obs_dim=29, act_dim=8, hidden=32):
    g = torch.Generator().manual_seed(123)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    data = dict(obs=r(T, n_global, obs_dim), act=r(T, n_global, act_dim), adv=r(T, n_global), ret=r(T, n_global), val=r(T, n_global), logp=-8 + r(T, n_global))
    w = dict(w1=0.3 * r(hidden, obs_dim), b1=0.1 * r(hidden), w2=0.3 * r(hidden, hidden), b2=0.1 * r(hidden), wa=0.1 * r(act_dim, hidden), ba=0.1 * r(act_dim),
             wv=0.3 * r(1, hidden), bv=0.1 * r(1), log_std=torch.full((act_dim,), -0.75, dtype=torch.float64))
    return data, w


def _ppo_updates(data, w, rank, world, epochs=2, minibatch=32):
    """Two epochs of optimiser steps on minibatches drawn from the GLOBAL flat index space; a rank only touches its own walkers."""
    from drloco_amd import collectives
    T, n_global = data['adv'].shape
    n_local = n_global // world
    sl = slice(rank * n_local, (rank + 1) * n_local)
    loc = {k: v[:, sl].reshape(T * n_local, *v.shape[2:]) for k, v in data.items()}
    params = [w[k].clone().requires_grad_(True) for k in w]
    wd = dict(zip(w.keys(), params))
    opt = torch.optim.Adam(params, lr=5e-4, eps=1e-5)
    bucket = collectives.FlatGradAllReducer(params)
    g = torch.Generator().manual_seed(7)               # the same permutations on every rank
    for s in range(epochs * (T * n_global // minibatch)):
        if s % (T * n_global // minibatch) == 0:
            perm = torch.randperm(T * n_global, generator=g)
        k = s % (T * n_global // minibatch)
        idx = collectives.shard_minibatch(perm[k * minibatch:(k + 1) * minibatch], n_global, rank, world)
        adv, n_mb = collectives.minibatch_adv_normalize(loc['adv'][idx])
        assert n_mb == minibatch
        loss = collectives.ppo_minibatch_loss(wd, loc['obs'][idx], loc['act'][idx], adv, loc['ret'][idx], loc['val'][idx], loc['logp'][idx], n_mb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        bucket.reduce()                                  # C2: one all-reduce of the flat gradient
        torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()
    return {k: p.detach().clone() for k, p in wd.items()}


def _grad_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    data, w = _ppo_problem()
    out = _ppo_updates(data, w, rank, world)
    torch.save(out, os.path.join(tmp, f'w{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_allreduce_matches_single_process(tmp_path):
    """C2 (north_star: 'RCCL all-reduce ... for the PPO advantage-normalisation statistics and gradient step'): two ranks x 8
    walkers end at the parameters of one rank x 16 walkers -- same minibatch indices, per-minibatch advantage statistics
    of the global minibatch, gradients summed through one flat bucket, then identical clipping and Adam steps."""
    sys.path.insert(0, ROOT)
    world, port = 2, _free_port()
    mp.spawn(_grad_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    data, w = _ppo_problem()
    want = _ppo_updates(data, w, 0, 1)
    moved = 0.0
    for r in range(world):
        got = torch.load(tmp_path / f'w{r}.pt')
        for k in want:
            assert float((got[k] - want[k]).abs().max()) < 1e-12, (r, k)
            moved = max(moved, float((want[k] - w[k]).abs().max()))
    assert moved > 1e-3                    # the steps did something
