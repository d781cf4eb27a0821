"""GPU tests of the exact launch shapes bench.py times (BASELINE configs[1]: 4096 walkers x 512 control steps, split workgroups,
dl_rollout_fixed launches of 448 + 64 steps, one dl_vecnormalize_steps call per run on the side stream), of the 512-steps-per-launch cap
of dl_rollout_fixed, of the 19-dof walker at the full per-GPU size, of dl_vecnormalize_steps against a direct numpy restatement, and of the
split workgroups' fault path (a hand-over that times out must raise, never continue on stale constraint rows).
Reference behaviour matched: MimicEnv.step drloco/mujoco/mimic_env.py:60-126 (incl. the exception path :86-91), VecNormalize.step_wait
(SB3 1.0)."""
import numpy as np
import pytest

from drloco_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return torch


def _bench_rollout(torch, model, refs, n, T, runs, acts, form):
    """One rollout the way bench.py's default configuration does it (form = 'bench') or one control step at a time (form = 'steps')."""
    from drloco_amd.rollout import HipRolloutBuffer
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    venv = HipVecEnv(num_envs=n, seed=1234, model=model, refs=refs)
    venv.set_split(True)
    vn = HipVecNormalize(venv)
    vn.multi_block_reduce = True                     # what enable_overlap() selects: same single-step reduction form on both sides
    buf = HipRolloutBuffer(T, n, 29, 8, torch.device('cuda'), gamma=0.995, gae_lambda=0.95)
    buf.actions.copy_(acts)
    vn.reset()
    last_obs = vn.norm_obs_t
    last_done = buf.next_starts
    last_done.fill_(1)
    buf.reset()
    buf.observations[0].copy_(last_obs); buf.episode_starts[0].copy_(last_done)
    if form == 'bench':
        vn.enable_overlap(chunk=max(runs))
        assert vn.batched_steps                     # split workgroups: the five-launch normalisation per run
        t0 = 0
        for r in runs:
            ts = range(t0, t0 + r)
            vn.steps_fixed(buf.actions[t0:t0 + r], [buf.observations[t + 1] if t + 1 < T else last_obs for t in ts], [buf.rewards[t] for t in ts], buf._starts[t0 + 1:t0 + r + 1])
            t0 += r
        vn.flush()
    else:
        for t in range(T):
            vn.step_tensors(buf.actions[t], obs_out=buf.observations[t + 1] if t + 1 < T else last_obs, rew_out=buf.rewards[t], done_out=buf._starts[t + 1])
    torch.cuda.synchronize()
    from drloco_amd import lib as L
    L.check(venv._lib.dl_fault_check(venv._h, None))
    out = dict(obs=buf.observations.cpu().clone(), rew=buf.rewards.cpu().clone(), starts=buf._starts.cpu().clone(), last=last_obs.cpu().clone(), state=venv.get_state(),
               raw_obs=torch.as_tensor(vn.get_original_obs()).clone(), raw_rew=torch.as_tensor(vn.get_original_reward()).clone(),
               om=np.array(vn.obs_rms.mean), ov=np.array(vn.obs_rms.var), oc=float(vn.obs_rms.count),
               rm=float(np.array(vn.ret_rms.mean)), rv=float(np.array(vn.ret_rms.var)), rc=float(vn.ret_rms.count),
               mon={k: venv.get_attr(k) for k in ('ep_len_smoothed', 'ep_ret_smoothed', 'mean_reward_smoothed', 'moved_distance')})
    venv.close()
    return out


@pytest.mark.parametrize('runs', [(512,), (448, 64)], ids=['one-launch-of-512', 'launches-of-448+64'])
def test_benchmark_launch_shapes_match_the_step_by_step_path(torch_cuda, model, refs, runs):
    """bench.py's headline configuration at its full size -- 4096 walkers x 512 control steps, split workgroups, ONE launch of 512 steps
    (the default schedule; round 2's 448 + 64 as well), the rollout's normalisations as one dl_vecnormalize_steps call -- against the same rollout taken one dl_step + dl_vecnormalize_step at a time:
    everything the simulation produces (episode boundaries, final walker state, cursors, walked distance, Monitor words, the raw
    observation / reward of the last step) is bit-identical; the normalised rollout-buffer contents agree to one float32 rounding of the
    normalisation and the moments to 1e-12 (the batched form sums with the start-of-run mean as its shift)."""
    torch = torch_cuda
    n, T = 4096, 512
    g = torch.Generator(device='cuda'); g.manual_seed(4321)
    acts = torch.clamp(0.5 * torch.randn(T, n, 8, device='cuda', generator=g), -1, 1)
    a = _bench_rollout(torch, model, refs, n, T, runs, acts, 'bench')
    b = _bench_rollout(torch, model, refs, n, T, runs, acts, 'steps')
    assert torch.equal(a['starts'], b['starts'])
    assert torch.equal(a['raw_obs'], b['raw_obs']) and torch.equal(a['raw_rew'], b['raw_rew'])
    for k in a['state']:
        assert np.array_equal(a['state'][k], b['state'][k]), k
    for k in a['mon']:
        assert a['mon'][k] == b['mon'][k], k
    ndone = int(a['starts'][1:].sum())
    assert 0 < ndone < n * T and ndone > 1000            # episodes end (and reset) inside both launches
    assert a['oc'] == b['oc'] and a['rc'] == b['rc']
    np.testing.assert_allclose(a['om'], b['om'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(a['ov'], b['ov'], rtol=1e-11)
    np.testing.assert_allclose([a['rm'], a['rv']], [b['rm'], b['rv']], rtol=1e-11)
    np.testing.assert_allclose(a['obs'].numpy(), b['obs'].numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(a['rew'].numpy(), b['rew'].numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(a['last'].numpy(), b['last'].numpy(), rtol=0, atol=2e-6)
    # the sanity checks bench.py applies to its own output hold here
    raw_ok = torch.isfinite(a['obs']).all() and torch.isfinite(a['rew']).all()
    assert raw_ok


@pytest.mark.parametrize('split', [False, True], ids=['one-wave', 'split-workgroups'])
def test_rollout_fixed_beyond_the_launch_cap(torch_cuda, model, refs, split):
    """dl_rollout_fixed with T = 600 > 512: the call is cut into launches of 512 + 88 control steps (EnvImpl::MULTI); outputs, final state
    and Monitor statistics are those of 600 single dl_step launches, bit for bit."""
    torch = torch_cuda
    from drloco_amd.vec_env import HipVecEnv
    n, T = 1000, 600
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    acts = torch.clamp(0.6 * torch.randn(T, n, 8, device='cuda', generator=g), -1, 1)
    a = HipVecEnv(num_envs=n, seed=5, model=model, refs=refs)
    b = HipVecEnv(num_envs=n, seed=5, model=model, refs=refs)
    if split:
        a.set_split(True); b.set_split(True)
    a.reset_tensors(); b.reset_tensors()
    obs = torch.zeros(T, n, 29, device='cuda'); rew = torch.zeros(T, n, device='cuda'); done = torch.zeros(T, n, dtype=torch.uint8, device='cuda')
    for t in range(T):
        a.step_tensors(acts[t], obs_out=obs[t], rew_out=rew[t], done_out=done[t])
    lib = a._lib
    from drloco_amd import lib as L
    L.check(lib.dl_profile(b._h, 1))
    obs2, rew2, done2 = b.rollout_fixed(acts)
    import ctypes as C
    ms, launches = C.c_double(), C.c_int32()
    L.check(lib.dl_profile_read(b._h, C.byref(ms), C.byref(launches)))
    assert launches.value == 2 and lib.dl_profile_steps(b._h) == T           # 512 + 88
    assert torch.equal(obs, obs2) and torch.equal(rew, rew2) and torch.equal(done, done2)
    assert done.sum() > 100
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    for name in ('ep_len_smoothed', 'ep_ret_smoothed', 'mean_reward_smoothed', 'moved_distance'):
        assert a.get_attr(name) == b.get_attr(name), name
    a.close(); b.close()


@pytest.mark.parametrize('lanes', [16, 'split'], ids=['one-wave-form', 'split-workgroups'])
def test_loco3d_full_size_properties(torch_cuda, lanes):
    """BASELINE configs[3]'s walker at the per-GPU size of the configs (4096 walkers): size-independent properties of a 24-step rollout on
    the synthetic loco3d table, in the one-wave form and in the look-ahead split workgroups (`bench.py --walker loco3d` times the latter)."""
    torch = torch_cuda
    from drloco_amd import mocap, models
    from drloco_amd.vec_env import HipVecEnv
    n, T = 4096, 24
    ang, vel = mocap.synthetic_loco3d(L=20000, seed=0)
    table = mocap.loco3d_table(ang, vel)
    mk = lambda num, base=0: HipVecEnv(models.WALKER_165CM, num_envs=num, seed=77, refs=table, env_index_base=base, lanes_per_walker=lanes)
    env = mk(n)
    assert env.split == (lanes == 'split')
    env.reset_tensors()
    assert (env.get_state()['cursor'][abi.DL_CUR_EPISODE] == 1).all()
    g = torch.Generator(device='cuda'); g.manual_seed(4321)
    acts = torch.clamp(0.5 * torch.randn(T, n, 13, device='cuda', generator=g), -1, 1)
    obs, rew, done = env.rollout_fixed(acts)
    torch.cuda.synchronize()
    assert obs.shape == (T, n, 47) and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    d = done.bool()
    assert (rew[d] == 0).all() and (rew[~d] >= 0.2).all() and (rew[~d] <= 1.2 + 1e-6).all()      # (far from the reference the imitation terms underflow: exactly the alive bonus)
    env2 = mk(n); env2.reset_tensors()
    obs2, rew2, done2 = env2.rollout_fixed(acts)
    assert torch.equal(obs, obs2) and torch.equal(rew, rew2) and torch.equal(done, done2)             # deterministic
    env3 = mk(1024, 2048); env3.reset_tensors()
    obs3, rew3, done3 = env3.rollout_fixed(acts[:, 2048:3072].contiguous())
    assert torch.equal(obs[:, 2048:3072], obs3) and torch.equal(rew[:, 2048:3072], rew3) and torch.equal(done[:, 2048:3072], done3)   # a shard == its columns
    assert np.array_equal(env.get_state()['cursor'][abi.DL_CUR_EPISODE] - 1, done.sum(0).cpu().numpy())
    # phase features: angle / pi in [-1, 1], radius >= 0
    assert (obs[..., 0:8:2].abs() <= 1 + 1e-6).all() and (obs[..., 1:8:2] >= 0).all()
    for e in (env, env2, env3):
        e.close()


def _numpy_vecnormalize_steps(obs, rew, done, gamma, eps, clip_o, clip_r, flags):
    """SB3 1.0 VecNormalize.step_wait, K times, exactly as written there (RunningMeanStd.update_from_moments with np.mean / np.var of the
    batch), float64."""
    K, B, D = obs.shape
    om, ov, oc = np.zeros(D), np.ones(D), 1e-4
    rm, rv, rc = 0.0, 1.0, 1e-4
    ret = np.zeros(B)
    obs_n, rew_n = np.zeros_like(obs, dtype=np.float64), np.zeros_like(rew, dtype=np.float64)

    def upd(m, v, c, x):
        bm, bv, bc = x.mean(0), x.var(0), x.shape[0]
        delta, tot = bm - m, c + bc
        M2 = v * c + bv * bc + delta * delta * c * bc / tot
        return m + delta * bc / tot, M2 / tot, tot

    for t in range(K):
        x = obs[t].astype(np.float64)
        if flags & 1:
            om, ov, oc = upd(om, ov, oc, x)
        obs_n[t] = np.clip((x - om) / np.sqrt(ov + eps), -clip_o, clip_o) if flags & 2 else x
        r = rew[t].astype(np.float64)
        if flags & 4:
            ret = ret * gamma + r
            rm, rv, rc = upd(rm, rv, rc, ret)
        rew_n[t] = np.clip(r / np.sqrt(rv + eps), -clip_r, clip_r) if flags & 8 else r
        if flags & 4:
            ret[done[t] != 0] = 0
    return obs_n, rew_n, (om, ov, oc, rm, rv, rc, ret)


@pytest.mark.parametrize('offset,scale', [(0.0, 1.0), (1e3, 1.0), (1e3, 1e-2)], ids=['centred', 'offset-1e3', 'offset-1e3-std-1e-2'])
def test_vecnormalize_steps_against_numpy(torch_cuda, offset, scale):
    """dl_vecnormalize_steps for a run as long as the benchmark's (K = 448) against a direct numpy restatement of K sequential
    VecNormalize.step_wait calls.  The batched form computes every step's batch variance as SS/B - (S/B)^2 with the moments at the START of
    the run as the shift of all K steps: an observation offset of 1e3 with unit (or 1e-2) spread is the stress case for that stale shift
    (first run: shift 0).  Bounds: moments to 1e-9 relative of the variance scale at offset 1e3 / std 1 (measured ~1e-11), 1e-5 at std 1e-2
    (cancellation of 1e6 against 1e-4 in float64: 1e-16 * 1e6 / 1e-4), float32 outputs to 2e-6 + that."""
    torch = torch_cuda
    import ctypes as C
    from drloco_amd import lib as L
    lib = L.load()
    K, B, D = 448, 512, 29
    rng = np.random.default_rng(5)
    drift = np.linspace(0, 0.5, K)[:, None, None]                                   # the batch mean moves during the run, as a learning walker's does
    obs = (offset + scale * (rng.standard_normal((K, B, D)) * np.linspace(0.5, 2, D) + drift)).astype(np.float32)
    rew = (0.2 + rng.random((K, B))).astype(np.float32)
    done = (rng.random((K, B)) < 0.01).astype(np.uint8)
    flags = 1 | 2 | 4 | 8
    ref_o, ref_r, (om, ov, oc, rm, rv, rc, ret) = _numpy_vecnormalize_steps(obs, rew, done, 0.99, 1e-8, 10.0, 10.0, flags)
    dev = 'cuda'
    t = lambda a, dt=None: torch.as_tensor(a, device=dev) if dt is None else torch.as_tensor(a, device=dev, dtype=dt)
    d_obs, d_rew, d_done = t(obs), t(rew), t(done)
    mean, var, cnt = torch.zeros(D, dtype=torch.float64, device=dev), torch.ones(D, dtype=torch.float64, device=dev), torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
    rmean, rvar, rcnt = torch.zeros(1, dtype=torch.float64, device=dev), torch.ones(1, dtype=torch.float64, device=dev), torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
    dret = torch.zeros(B, dtype=torch.float64, device=dev)
    out_o, out_r = torch.zeros(K, B, D, device=dev), torch.zeros(K, B, device=dev)
    po = torch.tensor([out_o[k].data_ptr() for k in range(K)], dtype=torch.int64).to(dev)
    pr = torch.tensor([out_r[k].data_ptr() for k in range(K)], dtype=torch.int64).to(dev)
    work = torch.empty(abi.vn_steps_workspace_bytes(K, B, D), dtype=torch.uint8, device=dev)
    vwork = torch.zeros(abi.vn_workspace_bytes(D) // 8, dtype=torch.float64, device=dev)
    st = abi.VecNormState()
    st.obs_mean, st.obs_var, st.obs_count = mean.data_ptr(), var.data_ptr(), cnt.data_ptr()
    st.ret, st.ret_mean, st.ret_var, st.ret_count = dret.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), rcnt.data_ptr()
    st.workspace = vwork.data_ptr()
    st.gamma, st.eps, st.clip_obs, st.clip_rew, st.flags = 0.99, 1e-8, 10.0, 10.0, flags
    p = lambda x: C.c_void_p(x.data_ptr())
    L.check(lib.dl_vecnormalize_steps(C.byref(st), K, p(d_obs), p(d_rew), p(d_done), B, D, p(po), p(pr), p(work), None))
    torch.cuda.synchronize()
    var_scale = (scale * np.linspace(0.5, 2, D)) ** 2
    tol = 1e-9 if scale >= 1.0 else 1e-5
    assert float(cnt) == oc and float(rcnt) == rc
    np.testing.assert_allclose(mean.cpu().numpy(), om, rtol=1e-13, atol=1e-10 * scale)
    err_v = np.abs(var.cpu().numpy() - ov) / var_scale
    assert err_v.max() < tol, err_v.max()
    np.testing.assert_allclose([float(rmean), float(rvar)], [rm, rv], rtol=1e-10)
    np.testing.assert_allclose(dret.cpu().numpy(), ret, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out_o.cpu().numpy(), ref_o, rtol=0, atol=2e-6 + 20 * tol)
    np.testing.assert_allclose(out_r.cpu().numpy(), ref_r, rtol=0, atol=2e-6)


def test_split_handover_timeout_raises(torch_cuda, model, refs):
    """A split workgroup whose hand-over runs out of its poll budget must not carry on with stale constraint rows: forced here by letting
    the constraint waves leave at once (poll budget 0) -- every dynamics wave's first request then times out.  Expected: the handle's fault
    word is set, the next C-ABI call returns DL_E_FAULT (a DrlocoFault in Python), every walker took the exception path of that step
    (mimic_env.py:86-91: reward 0, done, reset) and is counted as diverged; after dl_fault_clear + reset the handle steps normally again."""
    torch = torch_cuda
    import ctypes as C
    from drloco_amd import lib as L
    from drloco_amd.vec_env import HipVecEnv
    n = 200
    env = HipVecEnv(num_envs=n, seed=9, model=model, refs=refs, lanes_per_walker='split')
    env.debug_counters()                                     # enables the per-walker solver diagnostics ([3] = diverged steps)
    env.reset_tensors()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    acts = torch.clamp(0.5 * torch.randn(3, n, 8, device='cuda', generator=g), -1, 1)
    env.step_tensors(acts[0])
    torch.cuda.synchronize()
    code = C.c_int32(-1)
    L.check(env._lib.dl_fault_check(env._h, C.byref(code)))
    assert code.value == 0 and (env.rew > 0.2).all()
    L.check(env._lib.dl_debug_set_spin_limit(env._h, 64, 0))          # constraint waves: no patience at all; dynamics waves: 64 polls
    rew = torch.full((3, n), -7.0, device='cuda'); done = torch.zeros(3, n, dtype=torch.uint8, device='cuda'); obs = torch.zeros(3, n, 29, device='cuda')
    L.check(env._lib.dl_rollout_fixed(env._h, 3, C.c_void_p(acts.data_ptr()), C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), None))
    torch.cuda.synchronize()
    rc = env._lib.dl_fault_check(env._h, C.byref(code))
    assert rc == abi.DL_E_FAULT and (code.value & 1), (rc, code.value)
    assert b'constraint wave' in env._lib.dl_last_error()
    assert (rew == 0).all() and (done == 1).all() and torch.isfinite(obs).all()       # the exception path on every step of the launch, never stale physics
    with pytest.raises(L.DrlocoFault):
        env.step_tensors(acts[0])                             # sticky: nothing launches on a faulted handle
    with pytest.raises(L.DrlocoFault):
        env.get_state()
    L.check(env._lib.dl_fault_clear(env._h))
    cnt = env.debug_counters()
    assert (cnt[3] == 3).all()                                # three diverged steps per walker
    L.check(env._lib.dl_debug_set_spin_limit(env._h, -1, -1))
    env.reset_tensors()
    env.step_tensors(acts[1])
    torch.cuda.synchronize()
    L.check(env._lib.dl_fault_check(env._h, C.byref(code)))
    assert code.value == 0 and (env.rew > 0.2).all() and not env.done.any()
    env.close()


@pytest.mark.parametrize('lanes', [1, 16, 'split'], ids=['lane-per-walker', '16-lanes-per-walker', '16-lanes-split-workgroups'])
def test_outputs_beyond_float32_range_saturate(torch_cuda, model, refs, lanes):
    """A state that leaves float32's range in the LAST substep of a control step becomes an observation before any check sees it (MuJoCo checks
    at the start of an mj_step: the reference emits the value -- a finite float64 there -- and raises one control step later).  The float32
    kernels hand out +-3e38 instead of inf / NaN, so that VecNormalize's moments stay finite; the next step takes the exception path."""
    torch = torch_cuda
    from drloco_amd.vec_env import HipVecEnv
    n = 8
    env = HipVecEnv(num_envs=n, seed=4, model=model, refs=refs, lanes_per_walker=lanes)
    env.reset_tensors()
    st = env.get_state()
    q, v = st['qpos'].copy(), st['qvel'].copy()
    v[5, 0] = np.inf; v[7, 1] = -np.inf; v[9, 2] = np.nan; q[4, 3] = 1e30
    flags = np.zeros(n, np.int32); flags[:4] = 1                      # walkers 0..3: the step ends in the injected state
    env.debug_inject(qpos=q, qvel=v, flags=flags)
    a = torch.zeros(n, 8, device='cuda')
    obs, rew, done, _ = env.step_tensors(a)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    o = obs.cpu().numpy()
    big = np.float32(3.0e38)                                          # (which column: the policy mirrors observations on left steps)
    assert all((np.abs(o[i]) == big).sum() == 1 for i in range(3)) and (np.abs(o[3]) > 1e29).sum() == 1 and (np.abs(o[4:]) < 1e3).all()
    assert not done[:4].any()                                        # nothing is flagged in the step that produced the values ...
    obs, rew, done, _ = env.step_tensors(a)
    torch.cuda.synchronize()
    assert done[:4].all() and (rew[:4] == 0).all() and not done[4:].any() and torch.isfinite(obs).all()      # ... the next one takes the exception path
    env.close()
