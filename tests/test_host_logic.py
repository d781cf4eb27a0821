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
