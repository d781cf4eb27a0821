"""GPU tests of dl_collect_rollouts' persistent form (k_rollout_persistent: SB3's collect_rollouts -- drloco/train.py:110-133 with the policy of
drloco/custom/policies.py:13-51 -- as ONE launch: per control step the policy forward of a workgroup's own sixteen rows, MimicEnv.step of its
sixteen walkers and VecNormalize's moment update through one grid-wide exchange).
  * exact mode: bit-identical to the launch-per-step form (dl_rollout_policy: 3 launches per control step) when that form simulates with the
    split workgroups and reduces the moments in the blocked order the persistent kernel follows (dl_vecnormalize_step flag 32);
  * per-rollout moments (opt-in): every output is re-derived from a replay -- the recorded actions through dl_rollout_fixed give the raw
    observations / rewards, the start-of-rollout moments give their normalisation, dl_policy_forward on the recorded observations gives the
    actions, and numpy's one-batch RunningMeanStd.update gives the moments after the rollout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch


def _setup(torch, model, refs, n, T, seed=21, hidden=512, **vn_kw):
    from drloco_amd.policy import HipPolicy
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    venv = HipVecEnv(num_envs=n, seed=seed, model=model, refs=refs, ep_dur_max=60)        # episodes end by time-out (and by falls) inside the window
    venv.set_split(True)
    vn = HipVecNormalize(venv, **vn_kw)
    vn.blocked_reduce = True
    pol = HipPolicy(hidden=hidden, seed=4)
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
    vn.reset()
    return venv, vn, pol, buf, vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')


@pytest.mark.parametrize('n,T', [(4096, 40), (1000, 33), (16, 25), (5, 9), (8192, 12), (5000, 10), (12000, 8)],
                         ids=['full-size', 'ragged-63-workgroups', 'one-workgroup', 'partly-filled-workgroup', 'two-blocks-per-workgroup', 'ragged-two-blocks',
                              'three-blocks-crossing-group-boundaries'])
def test_persistent_rollout_is_the_launch_path_bit_for_bit(torch_cuda, model, refs, n, T):
    _persistent_vs_launches(torch_cuda, model, refs, n, T, 512)


@pytest.mark.parametrize('hidden', [256, 128])
@pytest.mark.parametrize('n,T', [(1000, 33), (5, 9), (6000, 8)], ids=['ragged-63-workgroups', 'partly-filled-workgroup', 'two-blocks-per-workgroup'])
def test_persistent_rollout_other_hidden_sizes(torch_cuda, model, refs, n, T, hidden):
    """The reference's hidden sizes are a config (drloco/config/hypers.py:98-99; default [512, 512]): 256 and 128 run through the same persistent kernel as eight
    waves x 2 / 1 tiles, bit-identical to the launch path (whose dl_policy_forward uses the same eight-wave form)."""
    _persistent_vs_launches(torch_cuda, model, refs, n, T, hidden)


def _persistent_vs_launches(torch, model, refs, n, T, hidden):
    res = []
    for persistent in (False, True):
        venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T, hidden=hidden)
        for rollout in range(2):                                # the second rollout starts from the first one's last observation and moments
            buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=persistent)
            assert buf.last_form == ('persistent' if persistent else 'launches')
        torch.cuda.synchronize()
        from drloco_amd import lib as L
        L.check(venv._lib.dl_fault_check(venv._h, None))
        st = venv.get_state()
        res.append(dict(observations=buf.observations.cpu().clone(), actions=buf.actions.cpu().clone(), values=buf.values.cpu().clone(), log_probs=buf.log_probs.cpu().clone(),
                        rewards=buf.rewards.cpu().clone(), episode_starts=buf.episode_starts.cpu().clone(), last_obs=last_obs.cpu().clone(), last_done=last_done.cpu().clone(),
                        om=torch.as_tensor(vn.obs_rms.mean), ov=torch.as_tensor(vn.obs_rms.var), oc=torch.tensor(vn.obs_rms.count), rm=torch.as_tensor(vn.ret_rms.mean),
                        rv=torch.as_tensor(vn.ret_rms.var), rc=torch.tensor(vn.ret_rms.count), ret=vn.ret.cpu().clone(), raw_obs=torch.as_tensor(vn.get_original_obs()),
                        raw_rew=torch.as_tensor(vn.get_original_reward()), counter=torch.tensor(pol.counter), **{'state_' + k: torch.as_tensor(v) for k, v in st.items()},
                        ep_len=torch.as_tensor(venv.get_attr('ep_len_smoothed')), moved=torch.as_tensor(venv.get_attr('moved_distance'))))
        venv.close()
    a, b = res
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].double() - b[k].double()).abs().max()))
    assert (a['episode_starts'].sum() > 0 or n < 100 or T < 20) and a['counter'] == 2 * T


@pytest.mark.parametrize('n', [300, 4500], ids=['one-block-per-workgroup', 'two-blocks-per-workgroup'])
@pytest.mark.parametrize('kw', [dict(training=False), dict(norm_reward=False), dict(norm_obs=False)], ids=['frozen', 'raw-rewards', 'raw-observations'])
def test_persistent_rollout_flag_combinations(torch_cuda, model, refs, kw, n):
    """VecNormalize's switches (load_env's evaluation env: training = False / norm_reward = False, drloco/common/utils.py:234-240) in both forms."""
    torch = torch_cuda
    T = 20 if n < 1000 else 8
    res = []
    for persistent in (False, True):
        venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T, **kw)
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=persistent)
        torch.cuda.synchronize()
        res.append([buf.observations.cpu().clone(), buf.actions.cpu().clone(), buf.rewards.cpu().clone(), buf.episode_starts.cpu().clone(), last_obs.cpu().clone(),
                    torch.as_tensor(vn.obs_rms.mean), torch.as_tensor(vn.ret_rms.var), torch.tensor(vn.obs_rms.count), vn.ret.cpu().clone()])
        venv.close()
    for x, y in zip(*res):
        assert torch.equal(x, y), kw


@pytest.mark.parametrize('n,hidden', [(1000, 512), (4096, 512), (6000, 512), (1000, 256), (4096, 128)],
                         ids=['one-block-per-workgroup', 'full-size', 'two-blocks-per-workgroup', 'hidden-256', 'hidden-128-full-size'])
def test_persistent_rollout_with_per_rollout_moments(torch_cuda, model, refs, n, hidden):
    torch = torch_cuda
    from drloco_amd.vec_env import HipVecEnv
    T = 48
    venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T, hidden=hidden)
    # give the moments a non-trivial start: one exact rollout first
    buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
    torch.cuda.synchronize()
    om0, ov0, oc0 = vn.obs_rms.mean.copy(), vn.obs_rms.var.copy(), vn.obs_rms.count
    rm0, rv0, rc0 = float(vn.ret_rms.mean), float(vn.ret_rms.var), vn.ret_rms.count
    ret0 = vn.ret.cpu().numpy().copy()
    st0 = venv.get_state()
    obs0 = last_obs.cpu().numpy().copy()
    counter0 = pol.counter
    buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments='per_rollout')
    torch.cuda.synchronize()
    from drloco_amd import lib as L
    L.check(venv._lib.dl_fault_check(venv._h, None))
    # (1) replay: the recorded actions from the recorded start state, one launch per control step with the same (split) step kernel
    rep = HipVecEnv(num_envs=n, seed=21, model=model, refs=refs, ep_dur_max=60)
    rep.set_split(True)
    rep.reset_tensors()
    rep.set_state(qpos=st0['qpos'], qvel=st0['qvel'], warm=st0['warm'], cursor=st0['cursor'], walked=st0['walked'])
    raw_o = torch.zeros(T, n, 29, device='cuda'); raw_r = torch.zeros(T, n, device='cuda'); dn = torch.zeros(T, n, dtype=torch.uint8, device='cuda')
    for t in range(T):
        rep.step_tensors(buf.actions[t], obs_out=raw_o[t], rew_out=raw_r[t], done_out=dn[t])
    torch.cuda.synchronize()
    assert torch.equal(dn[:-1], buf.episode_starts[1:]) and torch.equal(dn[-1], last_done)
    assert dn.sum() > 0
    # (2) the whole rollout is normalised with the moments at its start
    ro, rr = raw_o.cpu().numpy().astype(np.float64), raw_r.cpu().numpy().astype(np.float64)
    exp_obs = np.clip((ro - om0) / np.sqrt(ov0 + 1e-8), -10, 10).astype(np.float32)
    exp_rew = np.clip(rr / np.sqrt(rv0 + 1e-8), -10, 10).astype(np.float32)
    got_obs = np.concatenate([buf.observations[1:].cpu().numpy(), last_obs.cpu().numpy()[None]])
    assert np.array_equal(buf.observations[0].cpu().numpy(), obs0)
    np.testing.assert_allclose(got_obs, exp_obs, rtol=0, atol=1e-6)         # float64 formula on both sides; one float32 rounding
    np.testing.assert_allclose(buf.rewards.cpu().numpy(), exp_rew, rtol=0, atol=1e-6)
    # (3) the policy's outputs are those of dl_policy_forward on the recorded observations (same counter stream)
    from drloco_amd.policy import HipPolicy
    p2 = HipPolicy(hidden=hidden, seed=4)
    p2.counter = counter0
    for t in range(T):          # EVERY step (the 4x4x1 defect of round 4 showed in about one row in a thousand: dl_policy_pair.hpp)
        p2.counter = counter0 + t
        a, v, lp = p2.forward(buf.observations[t])
        assert torch.equal(a, buf.actions[t]) and torch.equal(v, buf.values[t]) and torch.equal(lp, buf.log_probs[t]), t
    # (4) moments after the rollout: RunningMeanStd.update fed all T x N samples as ONE batch; discounted returns advance per step
    def upd(m, v, c, x):
        bm, bv, bc = x.mean(0), x.var(0), x.shape[0]
        d, tot = bm - m, c + bc
        return m + d * bc / tot, (v * c + bv * bc + d * d * c * bc / tot) / tot, tot
    m1, v1, c1 = upd(om0, ov0, oc0, raw_o.cpu().numpy().astype(np.float64).reshape(T * n, 29))
    ret, rets = ret0.copy(), []
    dnh = dn.cpu().numpy()
    for t in range(T):
        ret = ret * 0.99 + rr[t]
        rets.append(ret.copy())
        ret[dnh[t] != 0] = 0
    rm1, rv1, rc1 = upd(rm0, rv0, rc0, np.concatenate(rets))
    np.testing.assert_allclose(vn.obs_rms.mean, m1, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(vn.obs_rms.var, v1, rtol=1e-10)
    assert vn.obs_rms.count == c1 and vn.ret_rms.count == rc1
    np.testing.assert_allclose([float(vn.ret_rms.mean), float(vn.ret_rms.var)], [rm1, rv1], rtol=1e-10)
    np.testing.assert_allclose(vn.ret.cpu().numpy(), ret, rtol=1e-12, atol=1e-12)
    venv.close(); rep.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('moments,n,T,R', [('per_rollout', 4096, 64, 2), ('per_rollout', 8192, 32, 2), ('per_step', 4096, 64, 2), ('per_step', 8192, 32, 2)],
                         ids=['pairs-4096', 'pairs-two-blocks', 'exact-4096', 'exact-two-blocks'])
def test_policy_outputs_of_the_persistent_kernels_soak(torch_cuda, model, refs, moments, n, T, R):
    """>= 500 k row-steps per case: EVERY action / value / log-probability a persistent rollout kernel wrote (k_rollout_pairs: inline-asm v_mfma_f32_4x4x1
    chains next to a busy partner wave; k_rollout_persistent: inline-asm v_mfma_f32_16x16x4 tiles) is, bit for bit, what the stand-alone dl_policy_forward
    gives for the recorded observation and counter.  The round-4 defect of the 4x4x1 chains (about one row in a thousand, only inside the rollout kernel
    with another wave on the SIMD) is what this soak would see; tools/check_mfma_overlap.py guards the instruction forms statically."""
    torch = torch_cuda
    from drloco_amd import lib as L
    from drloco_amd.policy import HipPolicy
    venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T, seed=33)
    p2 = HipPolicy(hidden=512, seed=4)
    rows = bad = 0
    for r in range(R):
        c0 = pol.counter
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments=moments)
        torch.cuda.synchronize()
        L.check(venv._lib.dl_fault_check(venv._h, None))
        assert buf.last_form == 'persistent'
        for t in range(T):
            p2.counter = c0 + t
            a, v, lp = p2.forward(buf.observations[t])
            wrong = (a != buf.actions[t]).any(1) | (v != buf.values[t]) | (lp != buf.log_probs[t])
            bad += int(wrong.sum())
            rows += n
    assert rows >= 500_000 and bad == 0, (rows, bad)
    assert torch.isfinite(buf.actions).all() and float(buf.actions.abs().max()) > 0
    venv.close()


@pytest.mark.parametrize('n,T,kw', [(1000, 33, {}), (5, 9, {}), (300, 12, dict(training=False)), (300, 12, dict(norm_reward=False))], ids=['ragged', 'tiny', 'frozen', 'raw-rewards'])
def test_per_step_sync_host_loop_is_the_launch_path_bit_for_bit(torch_cuda, model, refs, n, T, kw):
    """HipVecNormalize(sync='per_step') -- dl_vn_local_sums, (all-reduce), dl_vn_merge_sums, dl_vecnormalize_step | 64 in a host loop -- with ONE rank
    is dl_collect_rollouts' launch form with the blocked reduction order, bit for bit: the split of VecNormalize.step_wait into 'local sums' and
    'merge' changes nothing but where the collective goes (tests/test_gpu_distributed.py runs it with two ranks)."""
    torch = torch_cuda
    res = []
    for sync in ('per_rollout', 'per_step'):
        venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T, sync=sync, **kw)
        for rollout in range(2):
            buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=None if sync == 'per_step' else False)
            assert buf.last_form == ('host loop' if sync == 'per_step' else 'launches')
        torch.cuda.synchronize()
        res.append(dict(observations=buf.observations.cpu().clone(), actions=buf.actions.cpu().clone(), values=buf.values.cpu().clone(), log_probs=buf.log_probs.cpu().clone(),
                        rewards=buf.rewards.cpu().clone(), episode_starts=buf.episode_starts.cpu().clone(), last_obs=last_obs.cpu().clone(), last_done=last_done.cpu().clone(),
                        om=torch.as_tensor(vn.obs_rms.mean), ov=torch.as_tensor(vn.obs_rms.var), oc=torch.tensor(vn.obs_rms.count), rm=torch.as_tensor(vn.ret_rms.mean),
                        rv=torch.as_tensor(vn.ret_rms.var), rc=torch.tensor(vn.ret_rms.count), ret=vn.ret.cpu().clone(), counter=torch.tensor(pol.counter)))
        venv.close()
    a, b = res
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].double() - b[k].double()).abs().max()))
    assert a['counter'] == 2 * T and abs(float(a['oc']) - (1e-4 + 2 * T * n if kw.get('training', True) else 1e-4)) < 0.5


def test_per_step_sync_refusals(torch_cuda, model, refs):
    torch = torch_cuda
    from drloco_amd import lib as L
    venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, 64, 4, sync='per_step')
    for kw in (dict(persistent=True), dict(moments='per_rollout', persistent=True)):
        with pytest.raises(L.DrlocoError):
            buf.collect_rollouts(vn, pol, last_obs, last_done, **kw)
    with pytest.raises(ValueError):
        from drloco_amd.vec_env import HipVecNormalize
        HipVecNormalize(venv, sync='sometimes')
    # the C entry points check their arguments
    from drloco_amd import abi
    assert venv._lib.dl_vn_local_sums(None, None, None, None, None, 4, 29, 0.99, 5, None, None) == abi.DL_E_INVAL
    assert venv._lib.dl_vn_merge_sums(None, 8, None, None, None, None, None, None, 29, 5, None) == abi.DL_E_INVAL
    venv.close()


@pytest.mark.parametrize('n,T,hidden', [(4096, 12, 512), (1000, 17, 512), (5, 9, 512), (1000, 17, 256), (600, 11, 128)], ids=['full-size', 'ragged', 'partly-filled-workgroup', 'hidden-256', 'hidden-128'])
def test_persistent_rollout_19dof_walker(torch_cuda, n, T, hidden):
    """BASELINE config 4's walker with a policy in the loop as ONE launch per rollout (round 5: the look-ahead split workgroups exist for the 19-dof walker,
    sixteen of its regions + the moments fit a CU's LDS): k_rollout_persistent<TopoWalker165> in the exact mode is the launch-per-step form bit for bit (split step
    kernel, blocked reduction order), and the per-rollout relaxation gives the same buffers on both of its kernels (pair by pair on v_mfma_f32_4x4x1, workgroup
    tiles on 16x16x4)."""
    torch = torch_cuda
    from drloco_amd import lib as L, mocap, models
    from drloco_amd.policy import HipPolicy
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    ang, vel = mocap.synthetic_loco3d(L=6000, seed=0)
    table = mocap.loco3d_table(ang, vel)

    def setup():
        venv = HipVecEnv(models.WALKER_165CM, num_envs=n, seed=21, refs=table, ep_dur_max=10)          # episodes end (time-out) and reset inside the window
        venv.set_split(True)
        vn = HipVecNormalize(venv); vn.blocked_reduce = True
        pol = HipPolicy(obs_dim=47, act_dim=13, hidden=hidden, seed=4)
        buf = HipRolloutBuffer(T, n, 47, 13, torch.device('cuda'))
        vn.reset()
        return venv, vn, pol, buf, vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')

    def snapshot(venv, vn, buf, last_obs, last_done):
        torch.cuda.synchronize()
        L.check(venv._lib.dl_fault_check(venv._h, None))
        return dict(observations=buf.observations.cpu().clone(), actions=buf.actions.cpu().clone(), values=buf.values.cpu().clone(), log_probs=buf.log_probs.cpu().clone(),
                    rewards=buf.rewards.cpu().clone(), episode_starts=buf.episode_starts.cpu().clone(), last_obs=last_obs.cpu().clone(), last_done=last_done.cpu().clone(),
                    om=torch.as_tensor(vn.obs_rms.mean), ov=torch.as_tensor(vn.obs_rms.var), oc=torch.tensor(vn.obs_rms.count), rv=torch.as_tensor(vn.ret_rms.var),
                    ret=vn.ret.cpu().clone(), qpos=torch.as_tensor(venv.get_state()['qpos']), cursor=torch.as_tensor(venv.get_state()['cursor']))
    res = []
    for persistent in (False, True):
        venv, vn, pol, buf, last_obs, last_done = setup()
        for rollout in range(2):
            buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=persistent)
            assert buf.last_form == ('persistent' if persistent else 'launches')
        res.append(snapshot(venv, vn, buf, last_obs, last_done))
        venv.close()
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), (k, float((res[0][k].double() - res[1][k].double()).abs().max()))
    assert torch.isfinite(res[0]['observations']).all() and (res[0]['episode_starts'].sum() > 0 or n < 100)
    rel = []
    for tiles in (False, True):
        venv, vn, pol, buf, last_obs, last_done = setup()
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments='per_rollout', workgroup_tiles=tiles)
        rel.append(snapshot(venv, vn, buf, last_obs, last_done))
        venv.close()
    for k in rel[0]:
        if k in ('om', 'ov', 'rv'):
            np.testing.assert_allclose(rel[0][k].numpy(), rel[1][k].numpy(), rtol=1e-10, atol=1e-12)
        else:
            assert torch.equal(rel[0][k], rel[1][k]), k


def test_packaged_walking_policy_walks(torch_cuda):
    """drloco_amd/data/walking_policy.npz (examples/train_ppo.py, 8 M steps; what `bench.py --policy --checkpoint walking` runs): with its VecNormalize moments and sampled actions
    the walkers reach the 3000-step episode limit and walk > 15 m per episode (the reference's "stable walk": drloco/common/callback.py:336-369) -- on the persistent kernel.
    checkpoint.moment_seat: restore() puts the moments back bit for bit (every rollout of a FIXED policy starts from the checkpoint's statistics: free-running ones drift away
    from what the policy was trained on, EXPERIMENTS.md round 6)."""
    torch = torch_cuda
    from drloco_amd import checkpoint
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    n, T = 256, 512
    venv = HipVecEnv(num_envs=n, seed=7)
    venv.set_split(True)
    vn = HipVecNormalize(venv); vn.reset()
    pol, meta = checkpoint.load_walking_policy(vec_normalize=vn, seed=3)
    vn.norm_obs_t.copy_(venv.obs); vn._normalize_obs_inplace(vn.norm_obs_t)
    restore = checkpoint.moment_seat(vn)
    seat = [(r._mean.clone(), r._var.clone(), r._count.clone()) for r in (vn.obs_rms, vn.ret_rms)]
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'))
    last_obs, last_done = vn.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device='cuda')
    ends = 0
    for r in range(7):          # 3584 control steps: every walker past its first 3000-step episode
        restore()
        for (m, v, c), rms in zip(seat, (vn.obs_rms, vn.ret_rms)):
            assert torch.equal(rms._mean, m) and torch.equal(rms._var, v) and torch.equal(rms._count, c)
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
        assert buf.last_form == 'persistent'
        assert not torch.equal(vn.obs_rms._count, seat[0][2])          # the rollout advanced them
        ends += int(buf._starts[1:T + 1].sum().item())
    torch.cuda.synchronize()
    ep_len = np.asarray(venv.get_attr('ep_len_smoothed'), np.float64)
    moved = np.asarray(venv.get_attr('moved_distance'), np.float64)
    assert ep_len.mean() > 2500 and moved.mean() > 15.0, (ep_len.mean(), moved.mean(), ends)
    assert n <= ends < 2 * n, ends          # one time-limit end per walker and a few falls
    venv.close()


@pytest.mark.timeout(600)
def test_bench_checkpoint_line_keeps_walking(torch_cuda):
    """`bench.py --policy --checkpoint walking` (the contact-rich line of README / profiles/r06_table.txt) in a small shape: the line carries the walking self-check, hardly a walker
    falls in the last rollout, and VecNormalize's largest variance at the end is the checkpoint's advanced by ONE rollout (every rollout starts from the checkpoint's moments)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline', '--policy', '--checkpoint', 'walking', '--envs-per-gpu', '512', '--rollout-len', '256', '--steps', '6', '--warmup', '2'],
                       cwd=root, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    sc = d['self_check']
    assert sc['finite'] and 'walking' in sc and 'checkpoint' in sc['walking']['vecnormalize']
    assert sc['episodes_ended_last_rollout'] < 0.05 * 512, sc          # 2048 control steps: nobody at the 3000-step limit yet; a policy outside its input distribution falls within ~150
    assert sc['exception_path_steps'] == 0
    from drloco_amd import checkpoint
    with np.load(checkpoint.WALKING_POLICY) as z:
        v0 = float(z['obs_var'].max())
    assert 0.8 * v0 < sc['obs_rms_var_max'] < 1.2 * v0, (v0, sc['obs_rms_var_max'])


def test_persistent_form_refusals(torch_cuda, model, refs):
    torch = torch_cuda
    from drloco_amd import lib as L, mocap, models
    from drloco_amd.policy import HipPolicy
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    def attempt(venv, pol, **kw):
        vn = HipVecNormalize(venv)
        vn.reset()
        buf = HipRolloutBuffer(4, venv.num_envs, venv.obs_dim, venv.nu, torch.device('cuda'))
        buf.collect_rollouts(vn, pol, vn.norm_obs_t.clone(), torch.ones(venv.num_envs, dtype=torch.uint8, device='cuda'), **kw)
        torch.cuda.synchronize()
        return buf.last_form
    # other hidden sizes, float64, one lane per walker, too many walkers: the automatic choice is the launch form, an explicit request raises
    for venv, pol in ((HipVecEnv(num_envs=64, model=model, refs=refs), HipPolicy(hidden=64)), (HipVecEnv(num_envs=64, model=model, refs=refs, precision=64), HipPolicy(hidden=512)),
                      (HipVecEnv(num_envs=64, model=model, refs=refs, lanes_per_walker=1), HipPolicy(hidden=512)), (HipVecEnv(num_envs=32784, model=model, refs=refs), HipPolicy(hidden=512))):
        assert attempt(venv, pol) == 'launches'
        with pytest.raises(L.DrlocoError):
            attempt(venv, pol, persistent=True)
        with pytest.raises(L.DrlocoError):
            attempt(venv, pol, persistent=False, moments='per_rollout')
        venv.close()
    # the 19-dof walker: persistent since round 5 (one block of sixteen walkers per workgroup: <= 4096 walkers on 256 CUs), launch form beyond
    ang, vel = mocap.synthetic_loco3d(L=4000, seed=1)
    venv = HipVecEnv(models.WALKER_165CM, num_envs=32, refs=mocap.loco3d_table(ang, vel))
    assert attempt(venv, HipPolicy(obs_dim=47, act_dim=13, hidden=512)) == 'persistent'
    venv.close()
    venv = HipVecEnv(models.WALKER_165CM, num_envs=4112, refs=mocap.loco3d_table(ang, vel))
    assert attempt(venv, HipPolicy(obs_dim=47, act_dim=13, hidden=512)) == 'launches'
    with pytest.raises(L.DrlocoError):
        attempt(venv, HipPolicy(obs_dim=47, act_dim=13, hidden=512), persistent=True)
    venv.close()
    venv = HipVecEnv(num_envs=64, model=model, refs=refs)
    assert attempt(venv, HipPolicy(hidden=512)) == 'persistent'          # (dl_set_split need not be on: the persistent kernel IS the split form)
    venv.close()


def test_blocked_reduction_order_against_numpy(torch_cuda):
    """dl_vecnormalize_step with the blocked summation order (flag 32: blocks of 16 rows -> <= 8 groups -> total; the order the persistent
    rollout kernel follows) against numpy's RunningMeanStd.update for several consecutive steps, ragged batch sizes included: moments to
    1e-12, float32 outputs to one rounding."""
    torch = torch_cuda
    import ctypes as C
    from drloco_amd import abi, lib as L
    lib = L.load()
    rng = np.random.default_rng(3)
    for B, D in ((4096, 29), (1000, 29), (5, 29), (130, 47)):
        om, ov, oc, rm, rv, rc = np.zeros(D), np.ones(D), 1e-4, 0.0, 1.0, 1e-4
        ret = np.zeros(B)
        dev = 'cuda'
        mean, var, cnt = torch.zeros(D, dtype=torch.float64, device=dev), torch.ones(D, dtype=torch.float64, device=dev), torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
        rmean, rvar, rcnt = torch.zeros(1, dtype=torch.float64, device=dev), torch.ones(1, dtype=torch.float64, device=dev), torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
        dret = torch.zeros(B, dtype=torch.float64, device=dev)
        work = torch.zeros(abi.vn_workspace_bytes(D) // 8, dtype=torch.float64, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr())

        def upd(m, v, c, x):
            bm, bv, bc = x.mean(0), x.var(0), x.shape[0]
            d, tot = bm - m, c + bc
            return m + d * bc / tot, (v * c + bv * bc + d * d * c * bc / tot) / tot, tot
        for t in range(6):
            obs = (3.0 + rng.standard_normal((B, D)) * np.linspace(0.5, 2, D)).astype(np.float32)
            rew = (0.2 + rng.random(B)).astype(np.float32)
            done = (rng.random(B) < 0.05).astype(np.uint8)
            om, ov, oc = upd(om, ov, oc, obs.astype(np.float64))
            ret = ret * 0.99 + rew
            rm, rv, rc = upd(rm, rv, rc, ret)
            exp_o = np.clip((obs.astype(np.float64) - om) / np.sqrt(ov + 1e-8), -10, 10)
            exp_r = np.clip(rew / np.sqrt(rv + 1e-8), -10, 10)
            ret[done != 0] = 0
            d_obs, d_rew, d_done = torch.as_tensor(obs, device=dev), torch.as_tensor(rew, device=dev), torch.as_tensor(done, device=dev)
            out_o, out_r = torch.zeros(B, D, device=dev), torch.zeros(B, device=dev)
            L.check(lib.dl_vecnormalize_step(p(d_obs), p(d_rew), p(d_done), p(mean), p(var), p(cnt), p(dret), p(rmean), p(rvar), p(rcnt), B, D, 0.99, 1e-8, 10.0, 10.0,
                                             1 | 2 | 4 | 8 | 32, p(out_o), p(out_r), p(work), None))
            torch.cuda.synchronize()
            np.testing.assert_allclose(mean.cpu().numpy(), om, rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(var.cpu().numpy(), ov, rtol=1e-11)
            assert float(cnt) == oc and float(rcnt) == rc
            np.testing.assert_allclose([float(rmean), float(rvar)], [rm, rv], rtol=1e-11)
            np.testing.assert_allclose(dret.cpu().numpy(), ret, rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(out_o.cpu().numpy(), exp_o, rtol=0, atol=2e-6)
            np.testing.assert_allclose(out_r.cpu().numpy(), exp_r, rtol=0, atol=2e-6)


def test_grid_exchange_timeout_raises(torch_cuda, model, refs):
    """A workgroup of the persistent rollout kernel that gives up on the grid-wide exchange (forced: poll budget 0) sets the handle's fault
    word and stops writing; the next C-ABI call raises, dl_fault_clear + reset bring the handle back."""
    torch = torch_cuda
    import ctypes as C
    from drloco_amd import abi, lib as L
    venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, 200, 6)
    L.check(venv._lib.dl_debug_set_grid_spin(venv._h, 0))
    with pytest.raises(L.DrlocoFault):          # collect_rollouts waits for the persistent launch and looks at the fault word before anything reads the buffer
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
    code = C.c_int32(0)
    assert venv._lib.dl_fault_check(venv._h, C.byref(code)) == abi.DL_E_FAULT and (code.value & 4)
    assert b'grid-wide' in venv._lib.dl_last_error()
    with pytest.raises(L.DrlocoFault):
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
    L.check(venv._lib.dl_fault_clear(venv._h))
    L.check(venv._lib.dl_debug_set_grid_spin(venv._h, -1))
    vn.reset()
    last_obs = vn.norm_obs_t.clone(); last_done.fill_(1)
    buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)
    torch.cuda.synchronize()
    L.check(venv._lib.dl_fault_check(venv._h, None))
    assert torch.isfinite(buf.observations).all() and (buf.rewards >= 0).all()
    venv.close()


def test_pair_meeting_timeout_raises(torch_cuda, model, refs):
    """The per-rollout kernel's wave pairs meet through two mailbox counters (pair_sync).  A wave whose partner does not arrive within the poll budget
    (forced: budget 0) sets DL_FAULT_SRV_TIMEOUT and LEAVES the kernel -- it never carries on with stale data -- so the launch ends, the host sees the
    fault before it reads the buffer, and dl_fault_clear + reset bring the handle back (the header's contract for every bounded poll)."""
    torch = torch_cuda
    import ctypes as C
    from drloco_amd import abi, lib as L
    venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, 200, 6)
    L.check(venv._lib.dl_debug_set_grid_spin(venv._h, 0))
    with pytest.raises(L.DrlocoFault):
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments='per_rollout')
    code = C.c_int32(0)
    assert venv._lib.dl_fault_check(venv._h, C.byref(code)) == abi.DL_E_FAULT and (code.value & abi.DL_FAULT_SRV_TIMEOUT)
    L.check(venv._lib.dl_fault_clear(venv._h))
    L.check(venv._lib.dl_debug_set_grid_spin(venv._h, -1))
    vn.reset()
    last_obs = vn.norm_obs_t.clone(); last_done.fill_(1)
    buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments='per_rollout')
    torch.cuda.synchronize()
    L.check(venv._lib.dl_fault_check(venv._h, None))
    assert torch.isfinite(buf.observations).all() and (buf.rewards >= 0).all()
    venv.close()


@pytest.mark.parametrize('n', [1000, 6000], ids=['one-block-per-workgroup', 'two-blocks-per-workgroup'])
def test_per_rollout_moments_on_workgroup_tiles_is_the_pair_kernel(torch_cuda, model, refs, n):
    """DL_ROLLOUT_WORKGROUP_TILES: the per-rollout relaxation on the exact form's kernel (sixteen-row 16x16x4 policy tiles, the workgroup's pairs in
    lockstep) -- the selectable fallback for the pair-by-pair kernel (4x4x1 chains).  With frozen moments nothing couples the walkers, both policy forms
    give the bits of dl_policy_forward and both run the same step code: every buffer is identical; the merged moments differ only by the grouping of
    the float64 sums (a slot per block of sixteen instead of per pair)."""
    torch = torch_cuda
    T, res = 24, []
    for tiles in (False, True):
        venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T)
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True)                       # a non-trivial start for the moments
        buf.collect_rollouts(vn, pol, last_obs, last_done, persistent=True, moments='per_rollout', workgroup_tiles=tiles)
        torch.cuda.synchronize()
        from drloco_amd import lib as L
        L.check(venv._lib.dl_fault_check(venv._h, None))
        res.append((dict(observations=buf.observations.cpu().clone(), actions=buf.actions.cpu().clone(), values=buf.values.cpu().clone(), log_probs=buf.log_probs.cpu().clone(),
                         rewards=buf.rewards.cpu().clone(), episode_starts=buf.episode_starts.cpu().clone(), last_obs=last_obs.cpu().clone(), last_done=last_done.cpu().clone(),
                         ret=vn.ret.cpu().clone(), qpos=torch.as_tensor(venv.get_state()['qpos'])),
                    dict(om=vn.obs_rms.mean.copy(), ov=vn.obs_rms.var.copy(), oc=vn.obs_rms.count, rm=float(vn.ret_rms.mean), rv=float(vn.ret_rms.var), rc=vn.ret_rms.count)))
        venv.close()
    (a, ma), (b, mb) = res
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert ma['oc'] == mb['oc'] and ma['rc'] == mb['rc']
    np.testing.assert_allclose(ma['om'], mb['om'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ma['ov'], mb['ov'], rtol=1e-11)
    np.testing.assert_allclose([ma['rm'], ma['rv']], [mb['rm'], mb['rv']], rtol=1e-11)


def test_grid_exchange_timeout_falls_back_in_auto_mode(torch_cuda, model, refs):
    """persistent=None (what examples/train_ppo.py uses): when the persistent kernel's grid-wide exchange times out (forced here; in the field: another
    process or stream holding CUs) the SAME call clears the fault, restores the moments, resets the walkers and redoes the rollout with the launch
    form -- the learner never sees a half-written buffer -- and the buffer stays on the launch form afterwards."""
    torch = torch_cuda
    import warnings
    from drloco_amd import lib as L
    n, T = 200, 6
    venv, vn, pol, buf, last_obs, last_done = _setup(torch, model, refs, n, T)
    count0, counter0 = vn.obs_rms.count, pol.counter
    buf.observations.fill_(float('nan')); buf.rewards.fill_(float('nan'))
    L.check(venv._lib.dl_debug_set_grid_spin(venv._h, 0))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        buf.collect_rollouts(vn, pol, last_obs, last_done)
    torch.cuda.synchronize()
    assert any('launch form' in str(x.message) for x in w) and buf.last_form == 'launches'
    L.check(venv._lib.dl_fault_check(venv._h, None))                                   # the fault word is clear again
    assert torch.isfinite(buf.observations).all() and torch.isfinite(buf.rewards).all() and (buf.episode_starts[0] == 1).all()     # every slot was written by the redo
    assert vn.obs_rms.count == pytest.approx(count0 + n * T, abs=1e-9) and vn.ret_rms.count == pytest.approx(count0 + n * T, abs=1e-9)   # moments: restored, then T batches
    assert pol.counter == counter0 + T
    # the redone rollout is a regular launch-form rollout: replaying its recorded actions from the state the redo started from gives its rewards
    L.check(venv._lib.dl_debug_set_grid_spin(venv._h, -1))
    buf.collect_rollouts(vn, pol, last_obs, last_done)          # stays on the launch form
    assert buf.last_form == 'launches' and vn.obs_rms.count == pytest.approx(count0 + 2 * n * T, abs=1e-9)
    venv.close()


@pytest.mark.parametrize('hidden', [512, 256, 128])
def test_policy_forms_agree_bit_for_bit(torch_cuda, hidden):
    """dl_policy_forward picks its form by batch size (<= 4096 rows: the whole first-layer block staged in LDS, barrier-free hidden layer;
    above: the lean 23 KB form that fits next to env-step workgroups); rollouts add the packed weight layout.  Same order of arithmetic in
    all of them: the first 4096 rows of an 8192-row call (lean) equal a 4096-row call (whole) bit for bit."""
    torch = torch_cuda
    from drloco_amd.policy import HipPolicy
    pol = HipPolicy(hidden=hidden, seed=11)
    g = torch.Generator(device='cuda'); g.manual_seed(2)
    obs = torch.randn(8192, 29, device='cuda', generator=g)
    eps = torch.randn(8192, 8, device='cuda', generator=g)
    a8, v8, l8 = pol.forward(obs, eps=eps)
    a4, v4, l4 = pol.forward(obs[:4096].contiguous(), eps=eps[:4096].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(a8[:4096], a4) and torch.equal(v8[:4096], v4) and torch.equal(l8[:4096], l4)
    # HipPolicy.forward reads the packed copy of the weights (refreshed when torch's version counter of a weight tensor moves); the C-ABI's
    # dl_policy_forward reads torch's layout: same bits
    import ctypes as C
    from drloco_amd import lib as L
    from drloco_amd.vec_env import _ptr, _stream
    for n in (4096, 8192):
        a, v, lp = torch.empty(n, 8, device='cuda'), torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
        p = pol._params()
        L.check(pol._lib.dl_policy_forward(C.byref(p), _ptr(obs[:n].contiguous()), n, _ptr(eps[:n].contiguous()), pol.seed, 0, 0, 0, _ptr(a), _ptr(v), _ptr(lp), _stream()))
        torch.cuda.synchronize()
        assert torch.equal(a, a8[:n]) and torch.equal(v, v8[:n]) and torch.equal(lp, l8[:n]), n
    # an in-place update of a weight (what an optimiser step is) is seen by the next forward
    pol.w2.mul_(1.5)
    a9, _, _ = pol.forward(obs[:64].contiguous(), eps=eps[:64].contiguous())
    ref, _, _ = pol.torch_reference(obs[:64], eps[:64])
    torch.cuda.synchronize()
    assert not torch.equal(a9, a4[:64]) and torch.allclose(a9, ref, atol=2e-5)

@pytest.mark.parametrize('hidden', [512, 256, 128])
def test_policy_pair_form_is_the_forward_pass_bit_for_bit(torch_cuda, hidden):
    """dl_policy_forward_pair (four rows per wave pair on v_mfma_f32_4x4x1_16B_f32, single-k instructions issued in the order in which the 16x16x4 tiles of
    dl_policy_forward accumulate) gives the same bits: sampled actions with given draws and with the counter-based stream, deterministic actions, values,
    log-probabilities; ragged row counts."""
    torch = torch_cuda
    import ctypes as C
    from drloco_amd import lib as L
    from drloco_amd.policy import HipPolicy
    from drloco_amd.vec_env import _ptr, _stream
    pol = HipPolicy(hidden=hidden, seed=11)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for n in (4096, 1001, 6, 3):
        obs = 3.0 * torch.randn(n, 29, device='cuda', generator=g)
        eps = torch.randn(n, 8, device='cuda', generator=g)
        for mode in ('eps', 'stream', 'deterministic'):
            pol.counter = 7
            a0, v0, l0 = pol.forward(obs, eps=eps if mode == 'eps' else None, deterministic=mode == 'deterministic')
            a, v, lp = torch.empty(n, 8, device='cuda'), torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
            p = pol._params()
            L.check(pol._lib.dl_policy_forward_pair(C.byref(p), _ptr(pol._packed_weights()), _ptr(obs), n, _ptr(eps) if mode == 'eps' else None, pol.seed, 7, pol.index_base,
                                                    int(mode == 'deterministic'), _ptr(a), _ptr(v), _ptr(lp), _stream()))
            torch.cuda.synchronize()
            assert torch.equal(a, a0) and torch.equal(v, v0) and torch.equal(lp, l0), (n, mode, float((a - a0).abs().max()), float((v - v0).abs().max()))
