"""CPU tests of the host-side logic that does not need a device: MJCF parsing vs the baked
model, mocap conversion, descriptor plumbing, the kernels' source running on the host
(tests/host_emu) against the independent oracle."""
import os
import sys

import numpy as np
import pytest

from drloco_amd import abi, mjcf, mocap, models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_emu'))
REF_XML = '/root/reference/drloco/mujoco/xml/walker3d_flat_feet.xml'
REF_MAT = '/root/reference/mocaps/straight_walking/Trajecs_Constant_Speed_400Hz.mat'


@pytest.fixture(scope='module')
def emu():
    import emu as E
    E.lib()
    return E


def test_baked_model_constants(model):
    assert (model.nbody, model.nv, model.nu, model.ngeom, model.nsite, model.frame_skip) == (8, 14, 8, 7, 8, 5)
    assert abs(sum(model.body_mass[:8]) - 80.5) < 1e-12
    assert model.timestep == 0.001 and model.jnt_qpos0[2] == 1.08
    # thigh capsule runs from z=-0.05 down to z=-0.45: the geom frame's z axis points down
    assert model.geom_mat[1][8] == pytest.approx(-1.0)
    assert [model.act_dof[a] for a in range(8)] == list(range(6, 14))


@pytest.mark.skipif(not os.path.exists(REF_XML), reason='reference checkout not present')
def test_mjcf_parser_matches_baked_model(model):
    parsed = mjcf.parse_mjcf(REF_XML, frame_skip=5)
    assert bytes(parsed) == bytes(model)


@pytest.mark.skipif(not os.path.exists(REF_MAT), reason='reference checkout not present')
def test_mocap_conversion_is_reproducible(refs):
    t = mocap.convert_straight_walk_mat(REF_MAT)
    assert np.array_equal(t.table, refs.table) and np.array_equal(t.step_off, refs.step_off)
    assert np.array_equal(t.step_vel, refs.step_vel) and np.array_equal(t.step_is_left, refs.step_is_left)


def test_mjcf_rejects_unsupported(tmp_path):
    p = tmp_path / 'm.xml'
    p.write_text('<mujoco><compiler angle="degree" coordinate="local" inertiafromgeom="false"/><option integrator="RK4"/></mujoco>')
    with pytest.raises(ValueError):
        mjcf.parse_mjcf(str(p), 5)
    with pytest.raises(ValueError):
        models.make_model('NoSuchWalker')


@pytest.mark.parametrize('precision,tol', [(64, 1e-10), (32, 5e-3)])
def test_kernel_source_on_host_matches_oracle_forward(emu, oracle, model, refs, precision, tol):
    """drloco_amd/csrc/dl_core.hpp compiled for the host vs oracle/dl_oracle.c: two independent
    formulations of the same dynamics."""
    n = 192
    cfg = abi.default_config()
    rng = np.random.default_rng(0)
    q = np.array(model.jnt_qpos0[:14])[:, None] + 0.25 * rng.standard_normal((14, n)); q[2] = rng.uniform(0.85, 1.3, n)
    v = 1.5 * rng.standard_normal((14, n)); w = rng.standard_normal((14, n)); u = rng.uniform(-300, 300, (8, n))
    o = oracle.OracleEnv(model, refs, cfg, n); e = emu.EmuEnv(model, refs, cfg, n, precision)
    o.set_state(qpos=q, qvel=v, warm=w); e.set_state(qpos=q, qvel=v, warm=w)
    qa, nc, ne, ni = o.forward(u); qb, nc2, ne2, ni2 = e.forward(u)
    assert np.array_equal(nc, nc2) and np.array_equal(ne, ne2)
    err = np.abs(qa - qb) / (1 + np.abs(qa))
    assert err.max() < tol


def test_kernel_source_on_host_matches_oracle_rollout(emu, oracle, model, refs):
    n, T = 48, 110
    cfg = abi.default_config()
    o = oracle.OracleEnv(model, refs, cfg, n); e = emu.EmuEnv(model, refs, cfg, n, 64)
    np.testing.assert_allclose(e.reset(), o.reset(), atol=2e-6)
    rng = np.random.default_rng(1)
    nd = 0
    for t in range(T):
        a = np.clip(0.5 * rng.standard_normal((n, 8)), -1, 1).astype(np.float32)
        o1, r1, d1, t1, _ = o.step(a.astype(np.float64)); o2, r2, d2, t2, _ = e.step(a)
        assert np.array_equal(d1, d2)
        np.testing.assert_allclose(o2, o1, atol=5e-5, rtol=2e-6)
        np.testing.assert_allclose(r2, r1, atol=1e-6)
        nd += int(d1.sum())
    s1, s2 = o.get_state(), e.get_state()
    assert np.array_equal(s1['cursor'], s2['cursor'])
    np.testing.assert_allclose(s2['walked'], s1['walked'], rtol=1e-9, atol=1e-12)
    assert nd > 0


def test_config_defaults_follow_the_reference():
    c = abi.default_config()
    assert list(c.rew_weights) == [0.8, 0.2, 0.0] and c.alive_bonus == 0.2 and c.ep_dur_max == 3000
    assert c.mirror_policy == 1 and c.ctrl_freq == 200.0 and c.com_z_min == 0.5


REF_XML_165 = '/root/reference/drloco/mujoco/xml/walker_165cm_65kg.xml'


@pytest.mark.skipif(not os.path.exists(REF_XML_165), reason='reference checkout not present')
def test_mjcf_parser_matches_baked_165cm_model():
    m = models.make_model(models.WALKER_165CM)
    assert bytes(mjcf.parse_mjcf(REF_XML_165, frame_skip=10)) == bytes(m)
    assert (m.nv, m.nu, m.nbody, m.ngeom, m.frame_skip) == (19, 13, 9, 8, 10)
    assert abs(sum(m.body_mass[:9]) - 65.17) < 1e-9
    # actuator order differs from joint order for the lumbar joints (xml:81-84)
    assert [m.act_dof[a] for a in range(3)] == [7, 6, 8]


def test_kernel_source_on_host_matches_oracle_loco3d(emu, oracle):
    """Second topology (19 dof, boxes on pelvis/torso, negated axes) + the loco3d env logic."""
    m = models.make_model(models.WALKER_165CM)
    ang, vel = mocap.synthetic_loco3d(L=4000, seed=1)
    table = mocap.loco3d_table(ang, vel)
    cfg = abi.loco3d_config()
    n = 64
    o = oracle.OracleEnv(m, table, cfg, n); e = emu.EmuEnv(m, table, cfg, n, 64)
    rng = np.random.default_rng(0)
    q = np.array(m.jnt_qpos0[:19])[:, None] + 0.2 * rng.standard_normal((19, n)); q[2] = rng.uniform(0.75, 1.2, n)
    v = 1.5 * rng.standard_normal((19, n)); w = rng.standard_normal((19, n)); u = rng.uniform(-300, 300, (13, n))
    o.set_state(qpos=q, qvel=v, warm=w); e.set_state(qpos=q, qvel=v, warm=w)
    qa, nc, ne, _ = o.forward(u); qb, nc2, ne2, _ = e.forward(u)
    assert np.array_equal(nc, nc2) and np.array_equal(ne, ne2) and nc.max() >= 6
    assert (np.abs(qa - qb) / (1 + np.abs(qa))).max() < 1e-9
    np.testing.assert_allclose(e.reset(), o.reset(), atol=2e-6)
    for t in range(40):
        a = np.clip(0.5 * rng.standard_normal((n, 13)), -1, 1).astype(np.float32)
        o1, r1, d1, _, _ = o.step(a.astype(np.float64)); o2, r2, d2, _, _ = e.step(a)
        assert np.array_equal(d1, d2)
        np.testing.assert_allclose(o2, o1, atol=5e-5, rtol=2e-6)
        np.testing.assert_allclose(r2, r1, atol=1e-6)
    assert np.array_equal(o.get_state()['cursor'], e.get_state()['cursor'])


def test_checkpoint_formats_round_trip(tmp_path):
    """drloco_amd.checkpoint: the SB3 1.0 layouts of utils.save_model's two files (restated; SB3 is not installed,
    so the reader's tolerance for un-importable classes is what makes a reference-written file readable here)."""
    import pickle
    import types
    import zipfile
    import torch
    from drloco_amd import checkpoint as ck
    rng = np.random.default_rng(0)
    rms = lambda shape: types.SimpleNamespace(mean=rng.standard_normal(shape), var=rng.uniform(0.5, 2, shape), count=12345.5)
    vn = types.SimpleNamespace(num_envs=8, obs_rms=rms((29,)), ret_rms=rms(()), clip_obs=10.0, clip_reward=10.0, gamma=0.99,
                               epsilon=1e-8, training=True, norm_obs=True, norm_reward=False)
    p = str(tmp_path / 'env_17')
    ck.write_vecnormalize_sb3(vn, p)
    raw = open(p, 'rb').read()
    assert b'stable_baselines3.common.vec_env.vec_normalize' in raw and b'RunningMeanStd' in raw
    assert 'stable_baselines3' not in sys.modules                       # the placeholder modules are gone again
    with pytest.raises(Exception):
        pickle.loads(raw)                                               # plain pickle needs SB3 ...
    s = ck.read_vecnormalize(p)                                         # ... the tolerant reader does not
    assert np.array_equal(s['obs_rms']['mean'], vn.obs_rms.mean) and s['ret_rms']['var'] == vn.ret_rms.var
    assert s['obs_rms']['count'] == 12345.5 and s['norm_reward'] is False and s['gamma'] == 0.99
    # model.zip
    t = lambda *sh: torch.as_tensor(rng.standard_normal(sh), dtype=torch.float32)
    pol = types.SimpleNamespace(w1=t(64, 29), b1=t(64), w2=t(64, 64), b2=t(64), wa=t(8, 64), ba=t(8), wv=t(1, 64), bv=t(1), log_std=torch.full((8,), -0.75))
    z = str(tmp_path / 'model_17.zip')
    ck.write_policy_zip(pol, z)
    assert {'data', 'policy.pth', 'pytorch_variables.pth', '_stable_baselines3_version'} <= set(zipfile.ZipFile(z).namelist())
    back = ck.read_policy_zip(z)
    for k in ('w1', 'b1', 'w2', 'b2', 'wa', 'ba', 'wv', 'bv', 'log_std'):
        assert torch.equal(back[k], getattr(pol, k))
    # separate policy / value trunks cannot be mapped onto the shared-trunk kernel
    import io
    sd = torch.load(io.BytesIO(zipfile.ZipFile(z).read('policy.pth')), weights_only=True)
    sd['mlp_extractor.value_net.0.weight'] = sd['mlp_extractor.value_net.0.weight'] + 1
    buf = io.BytesIO(); torch.save(sd, buf)
    z2 = str(tmp_path / 'sep.zip')
    with zipfile.ZipFile(z2, 'w') as zz:
        zz.writestr('policy.pth', buf.getvalue())
    with pytest.raises(ValueError):
        ck.read_policy_zip(z2)


# ---------------------------------------------------------------------------------------------
# The 16-lanes-per-walker kernels (drloco_amd/csrc/dl_group.hpp, dl_group_env.hpp: the product's step kernel) compiled for
# the host with a wave emulated as 64 fibers (tests/host_emu/dl_group_emu.hpp) against the oracle.  Both walkers: the
# straight walker (14 lane dofs) and the 19-dof walker (16 lane dofs + 3 replicated root translations).
def _walker(which):
    if which == 'straight':
        return models.make_model(), mocap.RefTable.load(), abi.default_config()
    ang, vel = mocap.synthetic_loco3d(L=4000, seed=1)
    return models.make_model(models.WALKER_165CM), mocap.loco3d_table(ang, vel), abi.loco3d_config()


def _states(m, n, seed, zlo):
    rng = np.random.default_rng(seed)
    nv, nu = m.nv, m.nu
    q = np.array(m.jnt_qpos0[:nv])[:, None] + 0.22 * rng.standard_normal((nv, n)); q[2] = rng.uniform(zlo, zlo + 0.45, n)
    return q, 1.5 * rng.standard_normal((nv, n)), rng.standard_normal((nv, n)), rng.uniform(-300, 300, (nu, n))


@pytest.mark.parametrize('which,zlo', [('straight', 0.85), ('loco3d', 0.75)])
@pytest.mark.parametrize('precision,tol', [(64, 1e-10), (32, 5e-3)])
def test_group_kernel_source_on_host_forward(emu, oracle, which, zlo, precision, tol):
    m, table, cfg = _walker(which)
    n = 48
    q, v, w, u = _states(m, n, 3, zlo)
    o = oracle.OracleEnv(m, table, cfg, n); e = emu.EmuEnv(m, table, cfg, n, precision)
    o.set_state(qpos=q, qvel=v, warm=w); e.set_state(qpos=q, qvel=v, warm=w)
    qa, nc, ne, _ = o.forward(u); qb, nc2, ne2, _ = e.gforward(u)
    assert np.array_equal(nc, nc2) and np.array_equal(ne, ne2) and nc.max() >= 6
    err = np.abs(qa - qb) / (1 + np.abs(qa))
    assert err.max() < tol, err.max()
    # four waves per CU: the float32 build's LDS per wave (four walkers) times 4 fits the 160 KiB of a CU
    if precision == 32:
        assert 4 * ((e.glds_bytes() + 1279) // 1280 * 1280) <= 160 * 1024


@pytest.mark.parametrize('which,zlo', [('straight', 0.85), ('loco3d', 0.75)])
def test_group_kernel_source_on_host_randomization_and_push(emu, oracle, which, zlo):
    """dl_set_randomization / dl_set_push (BASELINE config 5, build-defined) for BOTH walkers: mass scale, floor friction and a
    push on the torso, float64 against the oracle."""
    m, table, cfg = _walker(which)
    n = 32
    rng = np.random.default_rng(9)
    ms = rng.uniform(0.8, 1.2, n).astype(np.float32); fr = rng.uniform(0.5, 1.1, n).astype(np.float32)
    push = np.zeros((n, 3), np.float32); k = rng.random(n) < 0.6
    ang = rng.uniform(0, 2 * np.pi, n); push[k, 0] = 50 * np.cos(ang[k]); push[k, 1] = 50 * np.sin(ang[k]); push[k, 2] = rng.uniform(-20, 20, k.sum())
    q, v, w, u = _states(m, n, 5, zlo)
    o = oracle.OracleEnv(m, table, cfg, n); e = emu.EmuEnv(m, table, cfg, n, 64)
    o.set_state(qpos=q, qvel=v, warm=w); e.set_state(qpos=q, qvel=v, warm=w)
    qa0, _, _, _ = o.forward(u)
    o.set_randomization(ms.astype(np.float64), fr.astype(np.float64), push.astype(np.float64)); e.set_randomization(ms, fr, push)
    qa, nc, ne, _ = o.forward(u); qb, nc2, ne2, _ = e.gforward(u)
    assert np.array_equal(nc, nc2) and np.array_equal(ne, ne2)
    assert (np.abs(qa - qb) / (1 + np.abs(qa))).max() < 1e-10
    assert np.abs(qa0 - qa).max() > 1.0          # the randomisation really changes the dynamics


@pytest.mark.parametrize('which,n,T', [('straight', 8, 14), ('loco3d', 8, 10)])
def test_group_kernel_source_on_host_rollout(emu, oracle, which, n, T):
    """Whole control steps of the product's step kernel (action map, RK4 x frame_skip, cursor, observation, reward, done,
    auto reset) on the host, float64, against the oracle; then the same steps again as ONE multi-step launch."""
    m, table, cfg = _walker(which)
    cfg.ep_dur_max = 9           # every walker times out inside the test: the in-kernel auto reset runs
    o = oracle.OracleEnv(m, table, cfg, n); e = emu.EmuEnv(m, table, cfg, n, 64); e2 = emu.EmuEnv(m, table, cfg, n, 64)
    np.testing.assert_allclose(e.reset(), o.reset(), atol=2e-6)
    st = o.get_state()
    e2.set_state(qpos=st['qpos'], qvel=st['qvel'], warm=e.get_state()['warm'], cursor=st['cursor'], walked=st['walked'])
    rng = np.random.default_rng(1)
    acts = np.clip(0.5 * rng.standard_normal((T, n, m.nu)), -1, 1).astype(np.float32)
    nd = 0
    single = []
    for t in range(T):
        o1, r1, d1, t1, _ = o.step(acts[t].astype(np.float64)); o2, r2, d2, t2, _ = e.gstep(acts[t])
        assert np.array_equal(d1, d2), t
        np.testing.assert_allclose(o2, o1, atol=5e-5, rtol=2e-6, err_msg=f't={t}')
        np.testing.assert_allclose(r2, r1, atol=1e-6)
        dm = d1.astype(bool)
        if dm.any():
            np.testing.assert_allclose(t2[dm], t1[dm], atol=5e-5, rtol=2e-6)
        nd += int(d1.sum())
        single.append((o2, r2, d2))
    assert nd >= n
    s1, s2 = o.get_state(), e.get_state()
    assert np.array_equal(s1['cursor'], s2['cursor'])
    np.testing.assert_allclose(s2['walked'], s1['walked'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(s2['qpos'], s1['qpos'], rtol=1e-7, atol=1e-8)
    # dl_rollout_fixed's long launches: bit-identical to single steps
    om, rm, dm_, _, _ = e2.gstep(acts)
    for t in range(T):
        assert np.array_equal(om[t], single[t][0]) and np.array_equal(rm[t], single[t][1]) and np.array_equal(dm_[t], single[t][2])


def test_group_kernel_source_on_host_ctrl_matches_G5(emu):
    """_rescale_actions + mirror_action of the step kernel itself against the reference's golden vector G5.  The C-ABI takes
    float32 actions and records float32 torques, so the comparison is exact up to those two roundings (2 ulp of float32);
    signs (incl. the -0.0 of a zero action), saturation and the left-step permutation must match exactly."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'G5_actions.npz'))
    m, table, cfg = _walker('straight')
    cfg.ep_dur_max = 10 ** 9
    n = len(g['actions'])
    e = emu.EmuEnv(m, table, cfg, 2 * n, 64)
    cur = np.zeros((abi.DL_CUR_WORDS, 2 * n), np.int32)
    cur[abi.DL_CUR_I_STEP] = np.concatenate([np.full(n, 4), np.full(n, 5)]); cur[abi.DL_CUR_READ_STEP] = cur[abi.DL_CUR_I_STEP]
    cur[abi.DL_CUR_POS] = 10; cur[abi.DL_CUR_COUNT] = 1
    e.set_state(cursor=cur)
    q_up = np.array(m.jnt_qpos0[:14])
    for k in range(2 * n):
        e.inject_state(k, q_up, np.zeros(14))
    out = e.gstep(np.concatenate([g['actions'], g['actions']]).astype(np.float32), return_ctrl=True)
    ctrl = out[-1].astype(np.float64)
    want = np.concatenate([g['rescaled'], g['mirrored']])
    np.testing.assert_allclose(ctrl, want, rtol=2.5e-7, atol=0)
    assert np.array_equal(np.signbit(ctrl), np.signbit(want)) and np.array_equal(np.abs(ctrl) == 300, np.abs(want) == 300)
    z = np.where((g['actions'] == 0).all(axis=1))[0][0]
    assert np.signbit(ctrl[z]).all()


def test_group_kernel_source_on_host_push_schedule(emu, oracle):
    """dl_set_push_schedule: the periodic push kept on the device, inside a multi-step launch, against the oracle stepped with
    the push switched on and off by hand (float64)."""
    m, table, cfg = _walker('straight')
    n, T, period, dur = 8, 12, 6, 2
    rng = np.random.default_rng(4)
    force = np.zeros((n, 3), np.float32); force[:, 0] = 50 * np.cos(np.arange(n)); force[:, 1] = 50 * np.sin(np.arange(n)); force[::3] = 0
    phase = rng.integers(0, period, n).astype(np.int32)
    o = oracle.OracleEnv(m, table, cfg, n); e = emu.EmuEnv(m, table, cfg, n, 64)
    np.testing.assert_allclose(e.reset(), o.reset(), atol=2e-6)
    e.set_push_schedule(force, phase, period, dur)
    acts = np.clip(0.5 * rng.standard_normal((T, n, 8)), -1, 1).astype(np.float32)
    om, rm, dm, _, _ = e.gstep(acts)                   # one launch of T control steps
    pushed_steps = 0
    for t in range(T):
        on = ((t + phase) % period) < dur
        pushed_steps += int((on & (np.abs(force).sum(1) > 0)).sum())
        o.set_randomization(xfrc=(force * on[:, None]).astype(np.float64))
        o1, r1, d1, _, _ = o.step(acts[t].astype(np.float64))
        assert np.array_equal(d1, dm[t]), t
        np.testing.assert_allclose(om[t], o1, atol=5e-5, rtol=2e-6, err_msg=f't={t}')
        np.testing.assert_allclose(rm[t], r1, atol=1e-6)
    assert pushed_steps > 10
    np.testing.assert_allclose(e.get_state()['qpos'], o.get_state()['qpos'], rtol=1e-7, atol=1e-8)


def test_bench_cpu_baselines_run(oracle):
    """bench.py's CPU legs (the oracle on one core, in the reference's SubprocVecEnv process structure, on many cores at once) on tiny
    samples: they fork worker processes before the benchmark touches the GPU, so they must keep working without one."""
    import bench
    one = bench.cpu_baseline(4, 3)
    sub = bench.cpu_baseline_subproc(2, 8)
    many = bench.cpu_baseline_all_cores(n_envs=4, n_steps=3, max_procs=2)
    for d, cores in ((one, 1), (sub, 3), (many, 2)):
        assert d['unit'] == 'env-steps/s' and d['value'] > 0 and d['cores'] == cores and isinstance(d['sample'], str)
    assert one['kind'] == 'port'
    assert len(bench.kernel_code_sha16()) == 16 and bench.kernel_code_sha16() == bench.kernel_code_sha16()


def test_bench_tapes_are_keyed_by_the_global_walker_index():
    """bench.py's action / value tapes (tape_normal): a counter-based generator keyed by (seed, stream, step, GLOBAL walker index, component) --
    a rank that draws only its own columns gets exactly the columns of the global tape, whatever the number of ranks; standard normal."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    full = bench.tape_normal(4321, 0, 32, 0, 96, 8, 'cpu')
    for world in (2, 3, 4):
        n = 96 // world
        for rank in range(world):
            assert torch.equal(bench.tape_normal(4321, 0, 32, rank * n, n, 8, 'cpu'), full[:, rank * n:(rank + 1) * n])
    assert not torch.equal(bench.tape_normal(4321, 1, 32, 0, 96, 8, 'cpu'), full) and not torch.equal(bench.tape_normal(4322, 0, 32, 0, 96, 8, 'cpu'), full)
    assert bench.tape_normal(4321, 2, 4, 5, 7, 0, 'cpu').shape == (4, 7)
    big = bench.tape_normal(7, 0, 64, 0, 4096, 8, 'cpu')
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1) < 5e-3 and float(big.abs().max()) < 6.5
    k = float(((big - big.mean()) ** 4).mean() / big.var() ** 2)
    assert abs(k - 3) < 0.05          # kurtosis of a normal


def test_bench_gpus_n_self_launch(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it: the parent -- before importing torch -- starts N ranks as a CHILD
    `python -m torch.distributed.run` on 127.0.0.1 with the same arguments and returns the child's exit code (never an exec: a process that
    touched the GPU must not be replaced)."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1'])
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    assert cmd[-7] == os.path.join(ROOT, 'bench.py') and cmd[-6:] == ['--gpus', '8', '--steps', '2', '--warmup', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # under a launcher (WORLD_SIZE set) nothing is started; a launcher that disagrees with --gpus is an error, not a silent resize
    monkeypatch.setenv('WORLD_SIZE', '4')
    seen.clear()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert not seen and 'disagree' in str(e.value.code)


@pytest.mark.parametrize('source', ['lane', 'group'])
def test_kernel_sources_on_host_follow_G15_quirk_Q4(emu, oracle, model, refs, source):
    """Quirk Q4 (adjust_COM_Z_pos re-anchors the data set in place; golden G15, generated by the reference) through BOTH kernel sources on the host, float64:
    the reference's COM-z values show up in the COM reward term of every step of all nine episodes, the per-step record (DevState::zacc) holds the offsets the
    reference left in its data set, and the Monitor's smoothed COM term follows."""
    from test_oracle_golden import g15_walk, load
    g = load('G15_q4_com_z.npz')
    cfg = abi.default_config(rew_weights=list(g['weights']))
    e = emu.EmuEnv(model, refs, cfg, 1, 64)
    if source == 'group':
        e.step = e.gstep
    zrow = lambda s: refs.table[2, refs.step_off[s]:refs.step_off[s + 1]]
    worst = dict(terms=0.0, rew=0.0)

    def per_reset(ep, obs0):
        st = e.get_state()
        np.testing.assert_allclose(st['qpos'][:, 0], g['ep_qpos0'][ep], rtol=0, atol=1e-12)
        np.testing.assert_allclose(obs0, g['ep_obs0'][ep], rtol=0, atol=2e-6)
        s = int(g['ep_read_step'][ep]); L = refs.step_len[s]
        assert st['cursor'][abi.DL_CUR_READ_STEP, 0] == s
        np.testing.assert_allclose(zrow(s)[[0, 1, L // 2, L - 1]] - e.get_ref_offsets()[s, 0], g['ep_zrow'][ep], rtol=0, atol=1e-12)

    def per_step(t, ep, obs, rew, done, term, terms):
        assert done == bool(g['done'][t]), t
        if not done:
            worst['terms'] = max(worst['terms'], np.abs(terms - g['terms'][t]).max()); worst['rew'] = max(worst['rew'], abs(rew - g['rew'][t]))
            np.testing.assert_allclose(obs, g['obs'][t], rtol=0, atol=2e-6)
    g15_walk(e, g, refs, per_step, per_reset)
    assert worst['terms'] < 2e-7 and worst['rew'] < 2e-7, worst          # float32 outputs of float64 arithmetic


@pytest.mark.parametrize('which', ['straight', 'loco3d'])
def test_strict_solver_follows_the_reference_solvers_decisions(emu, oracle, which):
    """dl_config.strict_solver: the 16-lane Newton solver takes [3P] mj_solNewton's decisions (start at the cheaper of warm start and qacc_smooth, exact line search every
    iteration, no early exit on an unchanged active set) -- the kernel source on the host, float64, against the oracle, which restates that solver: the SAME iteration count on
    (nearly) every random state, where the product's path (start at the warm start, full Newton step accepted when the active set stands) counts differently; the same minimiser
    either way."""
    m, table, cfg = _walker(which)
    n = 96
    q, v, w, u = _states(m, n, 4, 0.85 if which == 'straight' else 0.75)
    o = oracle.OracleEnv(m, table, cfg, n)
    o.set_state(qpos=q, qvel=v, warm=w)
    qa, nc, ne, ni = o.forward(u)
    import copy
    out = {}
    for strict in (0, 1):
        c2 = copy.copy(cfg); c2.strict_solver = strict
        e = emu.EmuEnv(m, table, c2, n, 64)
        e.set_state(qpos=q, qvel=v, warm=w)
        qb, nc2, ne2, ni2 = e.gforward(u)
        assert np.array_equal(nc, nc2) and np.array_equal(ne, ne2)
        assert (np.abs(qa - qb) / (1 + np.abs(qa))).max() < 1e-9          # the same minimiser
        out[strict] = ni2
        e.close()
    con = ne > 0
    assert con.sum() > n // 3
    same_strict, same_default = (out[1][con] == ni[con]).mean(), (out[0][con] == ni[con]).mean()
    assert same_strict >= 0.9, (same_strict, same_default, out[1][con][:20], ni[con][:20])
    assert same_strict > same_default + 0.2, (same_strict, same_default)


def test_device_code_hash_is_of_the_instructions():
    """lib.device_code_sha16: sha256 over .text + .rodata of the gfx950 code object in the library's offload bundle -- what profiles/*.json are stamped with.  Independent of the
    build directory and of -save-temps (checked in round 6 against a build of the same sources under /tmp: identical; over the whole .hip_fatbin it was not); the two code objects
    of the product differ (their hand-written DPP waits do)."""
    import os
    from drloco_amd import lib
    paths = [v['path'] for v in lib.VARIANTS.values() if os.path.exists(v['path'])]
    if len(paths) < 2:
        pytest.skip('both code objects are built by __graft_entry__.build()')
    a, b = (lib.device_code_sha16(p) for p in paths)
    assert len(a) == 16 and int(a, 16) >= 0 and a != b and a == lib.device_code_sha16(paths[0])
