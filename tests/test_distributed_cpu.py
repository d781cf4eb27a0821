"""World-size-2 gloo tests (CPU) of the N > 1 path.

The rollout shards by walker index with no data-path collective; the only exchange is the
all-reduce of the advantage-normalisation sums.  On CPU the same decomposition is exercised with
the oracle standing in for the kernels: (1) the RSI stream is keyed by the GLOBAL walker index, so
two shards of 8 walkers reproduce one 16-walker run bit for bit; (2) all-reducing the per-shard
[sum, sum^2, n] gives exactly the statistics of the concatenated batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    # the collective of the hot path: [sum, sum^2, n]
    sums = torch.tensor([adv.astype(np.float64).sum(), (adv.astype(np.float64) ** 2).sum(), adv.size], dtype=torch.float64)
    dist.all_reduce(sums)
    cnt, mean = sums[2].item(), sums[0].item() / sums[2].item()
    var = (sums[1].item() - cnt * mean * mean) / (cnt - 1)
    norm = (adv.astype(np.float64) - mean) / (np.sqrt(max(var, 0)) + 1e-8)
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), obs0=obs0, obs_last=obs_last, adv=adv, norm=norm,
             cursor=env.get_state()['cursor'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_rollout_and_advnorm_allreduce(tmp_path):
    world, port = 2, 29541 + os.getpid() % 500
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
    world, port = 2, 30041 + os.getpid() % 500
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
