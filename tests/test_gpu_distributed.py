"""bench.py's distributed branch on the one-GPU box: two fresh child ranks under torch.distributed.run (both on the one GPU, gloo as the
process-group backend -- RCCL needs one GPU per rank; the code path, the sharding and the exchanges are the ones the 8-GPU run takes) against
a single-process run of the same global walker count.  The reference's only parallelism is one process per environment
(drloco/common/utils.py:121-125); sharding contiguous walker ranges over ranks must not change what any walker does:
  * every rank's rollout -- episode boundaries, action tape, raw observations / rewards of every step, final cursors and positions -- equals
    the matching columns of the single-process run bit for bit (RSI draws and tapes are keyed by the GLOBAL walker index);
  * after the per-rollout exchange (collective C3) every rank holds the same VecNormalize moments, and they are the single process's to
    1e-10 (the ranks' per-step batches are halves of the single process's: same samples, another grouping of the merges);
  * with --vn-sync per_step (HipVecNormalize(sync='per_step'): the batch sums are all-reduced every control step) the ranks hold bitwise equal
    moments and follow the single process's rollout;
  * the JSON line reports world_size 2 and a positive whole-job value."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{') and '"metric"' in l]
    assert len(lines) == 1, p.stdout[-3000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize('extra', [[], ['--policy', '--rollout-form', 'launches'], ['--policy', '--rollout-form', 'launches', '--vn-sync', 'per_step']],
                         ids=['fixed-actions', 'policy-in-the-loop', 'policy-per-step-moments'])
def test_bench_two_ranks_match_one_process(tmp_path, extra):
    import torch
    assert torch.cuda.is_available()
    n, T = 512, 192
    common = ['--rollout-len', str(T), '--steps', '1', '--warmup', '0', '--no-cpu-baseline'] + extra
    env = dict(os.environ, DL_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = str(so.getsockname()[1])
    two = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', port,
                'bench.py', '--gpus', '2', '--envs-per-gpu', str(n), '--dump', str(tmp_path / 'two')] + common, env)
    one = _run([sys.executable, 'bench.py', '--gpus', '1', '--envs-per-gpu', str(2 * n), '--dump', str(tmp_path / 'one')] + common, dict(os.environ))
    assert two['distributed']['world_size'] == 2 and two['distributed']['backend'] == 'gloo' and two['n_gpus'] == 2
    assert two['value'] > 0 and one['value'] > 0 and two['self_check']['finite']
    ref = np.load(str(tmp_path / 'one') + '.rank0.npz')
    ranks = [np.load(str(tmp_path / 'two') + f'.rank{r}.npz') for r in range(2)]
    policy = bool(extra)
    per_step = '--vn-sync' in extra
    if per_step:
        # SB3's semantics across ranks: every control step's moment update uses the batch of BOTH ranks (one all-reduce of the batch sums per
        # control step), so each rank normalises -- and acts -- as the single process does.  The ranks hold bitwise equal moments at every step;
        # against the single process the sums are grouped differently (rounding of float64 sums), which a chaotic contact simulation amplifies
        # slowly: the first steps agree to float32 rounding, the moments after T steps to 1e-6.
        assert two['distributed']['vn_sync'] == 'per_step' and one['distributed']['vn_sync'] == 'per_step'
        for k in ('obs_mean', 'obs_var', 'obs_count', 'ret_mean', 'ret_var', 'ret_count'):
            assert np.array_equal(ranks[0][k], ranks[1][k]), k
        assert float(ranks[0]['obs_count']) == float(ref['obs_count']) and float(ranks[0]['ret_count']) == float(ref['ret_count'])
        for r, d in enumerate(ranks):
            cols = slice(r * n, (r + 1) * n)
            for t in range(8):
                np.testing.assert_allclose(d['observations'][t], ref['observations'][t][cols], rtol=0, atol=2e-5, err_msg=f'rank {r} step {t}')
                np.testing.assert_allclose(d['actions'][t], ref['actions'][t][cols], rtol=0, atol=2e-5)
            assert np.array_equal(d['starts'][:8], ref['starts'][:8, cols])
        np.testing.assert_allclose(ranks[0]['obs_mean'], ref['obs_mean'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ranks[0]['obs_var'], ref['obs_var'], rtol=1e-3)
        return
    for r, d in enumerate(ranks):
        cols = slice(r * n, (r + 1) * n)
        if not policy:
            # pre-generated actions: the simulation of a walker does not depend on the other walkers at all
            for k in ('starts', 'actions', 'values'):
                assert np.array_equal(d[k], ref[k][:, cols]), (r, k)
            for k in ('raw_obs', 'raw_rew'):
                assert np.array_equal(d[k], ref[k][:, cols]), (r, k)
            assert np.array_equal(d['cursor'], ref['cursor'][:, cols]) and np.array_equal(d['qpos'], ref['qpos'][:, cols])
            assert d['starts'][1:].sum() > 0
        else:
            # a policy in the loop sees observations normalised with its OWN rank's moments during a rollout (C3 merges between rollouts: the
            # documented relaxation), so trajectories separate after the first steps: the first observation / action rows still agree
            assert np.array_equal(d['observations'][0], ref['observations'][0][cols]) and np.array_equal(d['actions'][0], ref['actions'][0][cols])
            assert np.array_equal(d['starts'][:2], ref['starts'][:2, cols])
    # C3: after the exchange both ranks hold the same moments ...
    for k in ('obs_mean', 'obs_var', 'obs_count', 'ret_mean', 'ret_var', 'ret_count'):
        assert np.array_equal(ranks[0][k], ranks[1][k]), k
    assert float(ranks[0]['obs_count']) == float(ref['obs_count'])
    if not policy:
        # ... and they are the single process's (same samples; halves merged in another grouping)
        np.testing.assert_allclose(ranks[0]['obs_mean'], ref['obs_mean'], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(ranks[0]['obs_var'], ref['obs_var'], rtol=1e-10)
        np.testing.assert_allclose([float(ranks[0]['ret_mean']), float(ranks[0]['ret_var'])], [float(ref['ret_mean']), float(ref['ret_var'])], rtol=1e-10)


@pytest.mark.timeout(900)
def test_plain_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it and no WORLD_SIZE in the environment (how the driver starts N = 1): the parent
    starts the two ranks itself, forwards rank 0's single JSON line and exits with the children's code.  `ranks_seen` is an all-reduced 1 per
    rank: what the process group saw, not what the environment claimed."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['DL_BENCH_BACKEND'] = 'gloo'          # two ranks on the one GPU of this box; on a node the default (RCCL) runs
    out = _run([sys.executable, 'bench.py', '--gpus', '2', '--envs-per-gpu', '512', '--rollout-len', '64', '--steps', '1', '--warmup', '1'], env)
    assert out['n_gpus'] == 2 and out['distributed']['world_size'] == 2 and out['distributed']['ranks_seen'] == 2
    assert out['value'] > 0 and out['self_check']['finite'] and out['scaling'] == 'weak'
    assert 'cpu_baseline' not in out          # rank 0 at N = 1 only


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()          # (counting devices does not initialise the GPU on this image)
    except Exception:
        return 0


@pytest.mark.timeout(1200)
@pytest.mark.skipif(_gpu_count() < 2, reason='RCCL needs one GPU per rank: this box has fewer than two (the test enables itself on a multi-GPU node)')
@pytest.mark.parametrize('extra', [[], ['--policy']], ids=['fixed-actions', 'policy-in-the-loop'])
def test_rccl_two_ranks_match_one_process(tmp_path, extra):
    """The same comparison over RCCL (backend nccl, one GPU per rank, xGMI): costs nothing on a one-GPU box (skipped) and turns the first -m gpu run on a multi-GPU node into
    evidence that the collectives (C1 adv-norm sums, C3 moment merge) work over RCCL with more than one rank.  `python bench.py --gpus 2` starts its own ranks."""
    n, T = 512, 128
    common = ['--rollout-len', str(T), '--steps', '1', '--warmup', '0', '--no-cpu-baseline'] + extra
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'DL_BENCH_BACKEND')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    two = _run([sys.executable, 'bench.py', '--gpus', '2', '--envs-per-gpu', str(n), '--dump', str(tmp_path / 'two')] + common, env)
    one = _run([sys.executable, 'bench.py', '--gpus', '1', '--envs-per-gpu', str(2 * n), '--dump', str(tmp_path / 'one')] + common, env)
    d2 = two['distributed']
    assert d2['backend'] == 'nccl' and d2['world_size'] == 2 and d2['ranks_seen'] == 2 and two['n_gpus'] == 2
    assert len(d2['per_rank']['step_kernel_ms']) == 2 and min(d2['per_rank']['step_kernel_ms']) > 0
    ref = np.load(str(tmp_path / 'one') + '.rank0.npz')
    ranks = [np.load(str(tmp_path / 'two') + f'.rank{r}.npz') for r in range(2)]
    for r, d in enumerate(ranks):
        cols = slice(r * n, (r + 1) * n)
        if not extra:
            for k in ('starts', 'actions', 'values', 'raw_obs', 'raw_rew'):
                assert np.array_equal(d[k], ref[k][:, cols]), (r, k)
            assert np.array_equal(d['cursor'], ref['cursor'][:, cols]) and np.array_equal(d['qpos'], ref['qpos'][:, cols])
        else:
            assert np.array_equal(d['observations'][0], ref['observations'][0][cols]) and np.array_equal(d['actions'][0], ref['actions'][0][cols])
    for k in ('obs_mean', 'obs_var', 'obs_count', 'ret_mean', 'ret_var', 'ret_count'):
        assert np.array_equal(ranks[0][k], ranks[1][k]), k
    if not extra:
        np.testing.assert_allclose(ranks[0]['obs_mean'], ref['obs_mean'], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(ranks[0]['obs_var'], ref['obs_var'], rtol=1e-10)
