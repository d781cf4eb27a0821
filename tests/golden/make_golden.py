#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the *reference*
Python (read-only, /root/reference) in the build container.

This script only runs where /root/reference exists (never on the GPU box, never
from tests).  It injects dummy third-party modules (gym, mujoco_py, wandb,
seaborn, stable_baselines3 are not installed), selects the constant-speed mocap
file (the default ramp file is a missing blob, .MISSING_LARGE_BLOBS:2), builds a
MimicWalker3dEnv without MuJoCo (a fake `sim.data` and an injected
`do_simulation`), drives the reference's own methods and stores inputs and
outputs as .npz.  No reference source text is written anywhere.

Fixtures:
  G1_mocap_table.npz     converted table + lengths + left steps + step velocities
  G2_cursor_traces.npz   StraightWalkingTrajectories.next() traces
  G3_reward_obs.npz      reward terms and observations (+ mirroring)
  G4_step_traces.npz     MimicEnv.step() sequencing with injected dynamics
  G5_actions.npz         _rescale_actions + mirror_action
  G6_terminate_early.npz do_terminate_early truth table
  G7_monitor.npz         Monitor smoothing traces
  G11_mocap_options.npz  StraightWalkingTrajectories(mirror_refs=True) table + cursor trace; adapt_trajectories on the
                         synthetic loco3d table (`python tests/golden/make_golden.py mocap` regenerates only this one)
  G13_hip3d.npz          StraightWalking3dHipTrajectories.get_qpos / get_qvel (the 16-d padded rows) at a few cursors
                         (`python tests/golden/make_golden.py hip3d` regenerates only this one)
  G14_ramp_layout.npz    the reference's DEFAULT file layout (40 rows per step: Trajecs_Ramp_Slow_400Hz_EulerTrunkAdded.mat, a missing blob) on a
                         synthetic 250-step file in the reference's .mat schema: StraightWalkingTrajectories imported WITHOUT the
                         constant-speed patch (`python tests/golden/make_golden.py ramp` regenerates only this one)
  G15_q4_com_z.npz       quirk Q4: several episodes of ONE env whose resets re-anchor the COM-z row of the data set in place (reset_model ->
                         adjust_COM_Z_pos), incl. evaluation inits (`python tests/golden/make_golden.py q4` regenerates only this one)
  G10_policy_trunk.npz   CustomHiddenLayers (drloco/custom/policies.py:13-51): weights, inputs, latent outputs
                         (`python tests/golden/make_golden.py policy` regenerates only this one)

Usage:  python tests/golden/make_golden.py
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# stub environment for importing the reference
# --------------------------------------------------------------------------
class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        return _Dummy()


class _StubModule(types.ModuleType):
    """Module whose Capitalised attributes are dummy classes (so that
    `class MimicEnv(MujocoEnv, gym.utils.EzPickle)` can be built) and whose other
    attributes are sub-stubs."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name[:1].isupper():
            obj = type(name, (object,), {'__init__': lambda self, *a, **k: None})
        else:
            obj = _StubModule(self.__name__ + '.' + name)
            sys.modules[obj.__name__] = obj
        setattr(self, name, obj)
        return obj


def _install_stubs():
    for name in ['gym', 'gym.utils', 'gym.envs', 'gym.envs.mujoco', 'gym.envs.mujoco.mujoco_env',
                 'mujoco_py', 'mujoco_py.builder', 'wandb', 'seaborn',
                 'stable_baselines3', 'stable_baselines3.common',
                 'stable_baselines3.common.vec_env']:
        sys.modules[name] = _StubModule(name)
    # gym.Wrapper needs to behave like the real thing for Monitor
    class Wrapper:
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)
    sys.modules['gym'].Wrapper = Wrapper
    sys.modules['seaborn'].set = lambda *a, **k: None
    sys.modules['seaborn'].set_style = lambda *a, **k: None
    sys.modules['seaborn'].set_context = lambda *a, **k: None

    class MujocoException(Exception):
        pass
    sys.modules['mujoco_py.builder'].MujocoException = MujocoException
    sys.modules['mujoco_py'].builder = sys.modules['mujoco_py.builder']
    # `from collections import Iterable` (base_ref_trajecs.py:2) was removed in py3.10
    collections.Iterable = collections.abc.Iterable
    import matplotlib
    matplotlib.use('Agg', force=True)
    _use = matplotlib.use
    matplotlib.use = lambda *a, **k: None
    return MujocoException


def _import_reference():
    MujocoException = _install_stubs()
    sys.path.insert(0, REF)
    import importlib.util
    # straight_walk_trajecs computes its row constants at import time from which
    # file is selected (:24-27, :85-91); select the file that is present.
    path = os.path.join(REF, 'drloco/ref_trajecs/straight_walk_trajecs.py')
    src = open(path).read()
    assert 'PATH_REF_TRAJECS = PATH_SPEED_RAMP' in src
    src = src.replace('PATH_REF_TRAJECS = PATH_SPEED_RAMP', 'PATH_REF_TRAJECS = PATH_CONSTANT_SPEED')
    import drloco.ref_trajecs  # noqa: F401  (package)
    name = 'drloco.ref_trajecs.straight_walk_trajecs'
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src, path, 'exec'), mod.__dict__)
    sys.modules['drloco.ref_trajecs'].straight_walk_trajecs = mod
    from drloco.mujoco import mimic_walker3d
    from drloco.mujoco import monitor_wrapper
    from drloco.common import utils
    from drloco.config import hypers
    return mod, mimic_walker3d, monitor_wrapper, utils, hypers, MujocoException


def make_env(walker_mod, refs_mod):
    """Build a MimicWalker3dEnv without MuJoCo."""
    Env = walker_mod.MimicWalker3dEnv
    env = Env.__new__(Env)
    env.refs = refs_mod.StraightWalkingTrajectories(walker_mod.qpos_indices, walker_mod.qvel_indices)
    env.finished_init = True
    env._EVAL_MODEL = False
    env._FOLLOW_DESIRED_SPEED_PROFILE = False
    env._PLAYBACK_REF_TRAJECS = False
    env.pos_rew, env.vel_rew, env.com_rew = 0, 0, 0
    env.ep_dur = 0
    env.ep_rews = []
    env.mean_epret_smoothed = 0
    env.walked_distance = 0
    env.control_freq = 200
    env._frame_skip = 5
    data = types.SimpleNamespace(qpos=np.zeros(14), qvel=np.zeros(14),
                                 actuator_force=np.zeros(8), site_xpos=np.zeros((8, 3)))
    env.sim = types.SimpleNamespace(data=data)
    env.data = data
    env.action_space = types.SimpleNamespace(high=np.full(8, 300.0), low=np.full(8, -300.0))
    return env


def set_refs_cursor(refs, i_step, pos):
    """What get_random_init_state does (straight_walk_trajecs.py:460-474) with the
    two random draws replaced by the injected values."""
    refs._ep_dur = 0
    refs.dist = 0
    refs._i_step = i_step
    refs._step = refs.data[i_step]
    refs._qpos_full = refs._step
    refs._qvel_full = refs._step
    refs._trajec_len = refs._step.shape[1]
    refs._pos = pos


def make_loco3d_golden(mimic_env_mod, hypers, rng):
    """G9: MimicWalker165cm65kgEnv + Loco3dReferenceTrajectories driven on a SYNTHETIC table (the
    real loco3d_guoping.mat is a missing blob, .MISSING_LARGE_BLOBS:1) written to a temp dir in the
    reference's own .mat schema (angJoi, angDJoi, rowNameIK)."""
    import tempfile
    import scipy.io as spio
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from drloco_amd import mocap as my_mocap          # only the synthetic-data generator is used
    import drloco.config.config as cfgl
    L, seed = 3000, 3
    ang, vel = my_mocap.synthetic_loco3d(L=L, seed=seed)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'mocaps', 'loco3d'))
    spio.savemat(os.path.join(tmp, 'mocaps', 'loco3d', 'loco3d_guoping.mat'),
                 {'angJoi': ang, 'angDJoi': vel, 'rowNameIK': np.array([f'row{i}' for i in range(37)], dtype=object)})
    cfgl.ENV_ID = 'MimicWalker165cm65kg'
    cfgl.CTRL_FREQ = 100
    hypers.modification = hypers.MOD_CUSTOM_POLICY           # MOD_MIRR_POLICY cannot be used with loco3d refs
    import drloco.ref_trajecs.loco3d_trajecs as l3
    l3.get_project_path = lambda: tmp + '/'
    from drloco.mujoco import mimic_walker_165cm_65kg as w165
    Env = w165.MimicWalker165cm65kgEnv
    env = Env.__new__(Env)
    env.refs = l3.Loco3dReferenceTrajectories(w165.ref_trajecs_qpos_indices, w165.ref_trajecs_qvel_indices, w165.adaptations)
    env.finished_init = True
    env._EVAL_MODEL = False; env._FOLLOW_DESIRED_SPEED_PROFILE = False; env._PLAYBACK_REF_TRAJECS = False
    env.pos_rew, env.vel_rew, env.com_rew = 0, 0, 0
    env.ep_dur = 0; env.ep_rews = []; env.mean_epret_smoothed = 0; env.walked_distance = 0
    env.control_freq = 100; env._frame_skip = 10
    data = types.SimpleNamespace(qpos=np.zeros(19), qvel=np.zeros(19), actuator_force=np.zeros(13), site_xpos=np.zeros((8, 3)))
    env.sim = types.SimpleNamespace(data=data); env.data = data
    env.action_space = types.SimpleNamespace(high=np.full(13, 300.0), low=np.full(13, -300.0))
    refs = env.refs
    g = dict(L=L, seed=seed, table_checksum=np.array([ang.sum(), vel.sum(), np.abs(ang).sum()]),
             qpos_rows=np.array(w165.ref_trajecs_qpos_indices), stride=np.array(refs._increment))
    # cursor trace incl. the wrap (base_ref_trajecs.py:95-103)
    refs._pos = L - 300
    T = 200
    g['c_start'] = np.array(refs._pos)
    g['c_pos'] = np.zeros(T, np.int32); g['c_desvel'] = np.zeros((T, 2)); g['c_q'] = np.zeros((T, 19)); g['c_v'] = np.zeros((T, 19))
    for t in range(T):
        refs.next()
        g['c_pos'][t] = refs._pos
        g['c_desvel'][t] = refs.get_desired_walking_velocity_vector(False)
        g['c_q'][t] = refs.get_qpos(); g['c_v'][t] = refs.get_qvel()
    # observations / rewards around random cursor positions (incl. the last sample: empty mean -> nan)
    n = 96
    g['o_pos'] = np.zeros(n, np.int32); g['o_q'] = np.zeros((n, 19)); g['o_v'] = np.zeros((n, 19))
    g['o_obs'] = np.zeros((n, 47)); g['o_terms'] = np.zeros((n, 3)); g['o_imit'] = np.zeros(n)
    import warnings
    for k in range(n):
        p = int(rng.integers(0, L)) if k > 2 else [L - 1, L - 2, L - 120][k]
        refs._pos = p
        scale = [0.0, 0.01, 0.1, 0.4][k % 4]
        q = np.asarray(refs.get_qpos(), float) + scale * rng.standard_normal(19)
        v = np.asarray(refs.get_qvel(), float) + 8 * scale * rng.standard_normal(19)
        env.sim.data.qpos[:] = q; env.sim.data.qvel[:] = v
        g['o_pos'][k], g['o_q'][k], g['o_v'][k] = p, q, v
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            g['o_obs'][k] = env._get_obs()
        g['o_imit'][k] = env.get_imitation_reward()
        g['o_terms'][k] = [env.pos_rew, env.vel_rew, env.com_rew]
    # step() trace with injected dynamics (13 actions, no mirroring, wrap inside)
    T = 120
    refs._pos = L - 260
    env.ep_dur = 0; env.walked_distance = 0
    r2 = np.random.default_rng(11)
    sq = np.zeros((T, 19)); sv = np.zeros((T, 19))
    shadow_pos = refs._pos
    for t in range(T):
        shadow_pos += 5
        if shadow_pos >= L - 1: shadow_pos = 0
        save = refs._pos; refs._pos = shadow_pos
        sq[t] = np.asarray(refs.get_qpos(), float) + 0.05 * r2.standard_normal(19)
        sv[t] = np.asarray(refs.get_qvel(), float) + 0.5 * r2.standard_normal(19)
        refs._pos = save
    state = {'t': 0}; ctrls = []
    def do_simulation(ctrl, n_frames):
        ctrls.append(np.array(ctrl, float)); env.sim.data.qpos[:] = sq[state['t']]; env.sim.data.qvel[:] = sv[state['t']]; state['t'] += 1
    env.do_simulation = do_simulation
    acts = r2.uniform(-1.3, 1.3, (T, 13))
    g['s_start'] = np.array(refs._pos); g['s_actions'] = acts; g['s_q'] = sq; g['s_v'] = sv
    g['s_obs'] = np.zeros((T, 47)); g['s_rew'] = np.zeros(T); g['s_done'] = np.zeros(T, np.int32); g['s_ctrl'] = np.zeros((T, 13))
    g['s_walked'] = np.zeros(T); g['s_pos'] = np.zeros(T, np.int32)
    for t in range(T):
        o, rew, done, _ = env.step(acts[t])
        g['s_obs'][t], g['s_rew'][t], g['s_done'][t], g['s_ctrl'][t] = o, rew, done, ctrls[-1]
        g['s_walked'][t], g['s_pos'][t] = env.walked_distance, refs._pos
        assert not done
    np.savez_compressed(os.path.join(OUT, 'G9_loco3d.npz'), **g)


def make_policy_golden():
    """CustomHiddenLayers of the reference with a small hidden size (the fixture stays a few KB): the trunk that
    policy_net and value_net SHARE (both Sequentials are built from the same layer objects, policies.py:33-41)."""
    _install_stubs()
    sys.modules['stable_baselines3.common.policies'] = _StubModule('stable_baselines3.common.policies')
    sys.path.insert(0, REF)
    import torch as th
    from drloco.config import hypers
    hypers.hid_layer_sizes = [64, 64]
    hypers.activation_fns = [th.nn.Tanh] * 2
    from drloco.custom import policies
    th.manual_seed(0)
    net = policies.CustomHiddenLayers(29)
    assert net.policy_net[0] is net.value_net[0] and net.policy_net[2] is net.value_net[2]       # shared parameters
    x = th.randn(37, 29)
    with th.no_grad():
        lat_pi, lat_vf = net(x)
    assert th.equal(lat_pi, lat_vf)
    np.savez_compressed(os.path.join(OUT, 'G10_policy_trunk.npz'), x=x.numpy(), latent=lat_pi.numpy(),
                        w1=net.policy_net[0].weight.detach().numpy(), b1=net.policy_net[0].bias.detach().numpy(),
                        w2=net.policy_net[2].weight.detach().numpy(), b2=net.policy_net[2].bias.detach().numpy(),
                        shared=np.array(1), latent_dim_pi=np.array(net.latent_dim_pi), latent_dim_vf=np.array(net.latent_dim_vf))
    print('G10_policy_trunk.npz written')


def make_mocap_options_golden():
    """G11: the two load-time options of the reference-trajectory classes that G1/G9 do not exercise."""
    import tempfile
    import scipy.io as spio
    refs_mod, walker_mod, monitor_mod, utils, hypers, _ = _import_reference()
    qpos_rows, qvel_rows = list(walker_mod.qpos_indices), list(walker_mod.qvel_indices)
    refs = refs_mod.StraightWalkingTrajectories(qpos_rows, qvel_rows, mirror_refs=True)
    rows = qpos_rows + qvel_rows
    full = [np.asarray(s[rows, :], dtype=np.float64) for s in refs.data]
    # the first six steps in full, every step by per-row sums (keeps the fixture small)
    g = dict(m_table6=np.concatenate(full[:6], axis=1), m_rowsum=np.stack([f.sum(axis=1) for f in full]),
             m_rowabs=np.stack([np.abs(f).sum(axis=1) for f in full]),
             m_step_len=np.array([s.shape[1] for s in refs.data], dtype=np.int32),
             m_left=np.array(refs.left_step_indices, dtype=np.int32),
             m_step_vel=np.asarray(refs.step_velocities, dtype=np.float64))
    # a cursor trace across a right -> (mirrored) left rollover
    set_refs_cursor(refs, 4, 200)
    T = 120
    g['m_start'] = np.array([4, 200])
    g['m_q'] = np.zeros((T, 14)); g['m_v'] = np.zeros((T, 14)); g['m_cur'] = np.zeros((T, 2), np.int32)
    for t in range(T):
        refs.next()
        g['m_q'][t], g['m_v'][t] = np.asarray(refs.get_qpos(), float), np.asarray(refs.get_qvel(), float)
        g['m_cur'][t] = refs._i_step, refs._pos
    # adapt_trajectories through Loco3dReferenceTrajectories on a small synthetic table in the reference's .mat schema
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from drloco_amd import mocap as my_mocap
    import drloco.config.config as cfgl
    L, seed = 500, 5
    ang, vel = my_mocap.synthetic_loco3d(L=L, seed=seed)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'mocaps', 'loco3d'))
    spio.savemat(os.path.join(tmp, 'mocaps', 'loco3d', 'loco3d_guoping.mat'),
                 {'angJoi': ang, 'angDJoi': vel, 'rowNameIK': np.array([f'row{i}' for i in range(37)], dtype=object)})
    cfgl.CTRL_FREQ = 100
    import drloco.ref_trajecs.loco3d_trajecs as l3
    l3.get_project_path = lambda: tmp + '/'
    from drloco.mujoco import mimic_walker_165cm_65kg as w165
    adapt = {6: 1.1, 9: 0.9, 13: 1.1, 16: 0.9, 4: 0.95}
    r3 = l3.Loco3dReferenceTrajectories(w165.ref_trajecs_qpos_indices, w165.ref_trajecs_qvel_indices, adapt)
    g['a_L'], g['a_seed'] = np.array(L), np.array(seed)
    g['a_rows'], g['a_scalars'] = np.array(list(adapt.keys())), np.array(list(adapt.values()))
    g['a_q'] = np.asarray(r3._qpos_full[w165.ref_trajecs_qpos_indices, :], dtype=np.float64)
    g['a_v'] = np.asarray(r3._qvel_full[w165.ref_trajecs_qvel_indices, :], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'G11_mocap_options.npz'), **g)
    print('wrote G11_mocap_options.npz')


RAMP_STEPS, RAMP_SEED = 250, 14


def make_ramp_golden():
    """G14: the 40-row speed-ramp layout (straight_walk_trajecs.py:22-27,52-54,85-91: GRF rows 35-36, trunk Euler rows 37-39).  The real file is
    a missing blob; a synthetic file with the same schema (drloco_amd.mocap.synthetic_straight_walk -> write_straight_walk_mat, every sample a
    1 x 1 cell as in the reference's files) is written where the UNPATCHED reference module looks for its default file."""
    import tempfile
    _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from drloco_amd import mocap as my_mocap          # only the synthetic-data generator / writer are used
    import drloco.ref_trajecs.straight_walk_trajecs as refs_mod          # as the reference ships it: PATH_REF_TRAJECS = PATH_SPEED_RAMP
    assert not refs_mod._is_constant_speed and (refs_mod.GRF_R, refs_mod.TRUNK_ROT_X, refs_mod.TRUNK_ROT_Z) == (35, 37, 39)
    from drloco.mujoco import mimic_walker3d as walker_mod
    qpos_rows, qvel_rows = list(walker_mod.qpos_indices), list(walker_mod.qvel_indices)
    assert qpos_rows[3:6] == [37, 38, 39]
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'mocaps', 'straight_walking'))
    steps = my_mocap.synthetic_straight_walk(n_steps=RAMP_STEPS, seed=RAMP_SEED, n_rows=40)
    my_mocap.write_straight_walk_mat(os.path.join(tmp, refs_mod.PATH_SPEED_RAMP), steps, nested=True)
    refs_mod.get_project_path = lambda: tmp + '/'
    new_refs = lambda: refs_mod.StraightWalkingTrajectories(qpos_rows, qvel_rows)
    refs = new_refs()
    assert len(refs.data) == RAMP_STEPS and refs.data[0].shape[0] == 40
    rows = qpos_rows + qvel_rows
    full = [np.asarray(s[rows, :], dtype=np.float64) for s in refs.data]
    g = dict(n_steps=np.array(RAMP_STEPS), seed=np.array(RAMP_SEED), qpos_rows=np.array(qpos_rows), qvel_rows=np.array(qvel_rows),
             table_head=np.concatenate(full[:3], axis=1), table_tail=np.concatenate(full[-2:], axis=1),
             rowsum=np.stack([f.sum(axis=1) for f in full]), rowabs=np.stack([np.abs(f).sum(axis=1) for f in full]),
             step_len=np.array([s.shape[1] for s in refs.data], dtype=np.int32),
             left_step_indices=np.array(refs.left_step_indices, dtype=np.int32),
             step_velocities=np.asarray(refs.step_velocities, dtype=np.float64))
    # cursor traces through changing speeds, with count_steps_same_vel moving (Q2) and the wrap after the last step (:330-333)
    starts = [(0, 10, 1), (100, 5, 1), (247, 30, 1), (248, 0, 40), (37, 50, 20), (249, 3, 300)]
    T = 260
    tr = {k: np.zeros((len(starts), T)) for k in ['i_step', 'pos', 'len', 'count_same_vel', 'phase', 'desvel', 'is_left']}
    tr_q, tr_v = np.zeros((len(starts), T, 14)), np.zeros((len(starts), T, 14))
    for k, (i0, p0, c0) in enumerate(starts):
        refs = new_refs()
        refs.count_steps_same_vel = c0
        set_refs_cursor(refs, i0, p0)
        for t in range(T):
            refs.next()
            tr['i_step'][k, t], tr['pos'][k, t], tr['len'][k, t] = refs._i_step, refs._pos, refs._trajec_len
            tr['count_same_vel'][k, t] = refs.count_steps_same_vel
            tr['phase'][k, t] = refs.get_phase_variable()
            tr['desvel'][k, t] = refs.get_desired_walking_velocity_vector(False)[0]
            tr['is_left'][k, t] = refs.is_step_left()
            tr_q[k, t], tr_v[k, t] = np.asarray(refs.get_qpos(), float), np.asarray(refs.get_qvel(), float)
    # (Q2: the counter grows with the step index, so the desired velocity of a trace only moves at the wrap after the last step)
    assert len(set(tr['desvel'].reshape(-1).tolist())) > 4 and len(set(tr['desvel'][3].tolist())) > 1
    np.savez_compressed(os.path.join(OUT, 'G14_ramp_layout.npz'), starts=np.array(starts), ref_qpos=tr_q, ref_qvel=tr_v, **tr, **g)
    print('wrote G14_ramp_layout.npz')


def make_hip3d_golden():
    """G13: the trajectory class with frontal hip rows added (straight_walk_hip3d_trajecs.py:8-19; no env in env_map builds it)."""
    import contextlib, io
    refs_mod, walker_mod, monitor_mod, utils, hypers, _ = _import_reference()
    import drloco.ref_trajecs.straight_walk_hip3d_trajecs as h3
    qpos_rows, qvel_rows = list(walker_mod.qpos_indices), list(walker_mod.qvel_indices)
    refs = h3.StraightWalking3dHipTrajectories(qpos_rows, qvel_rows)
    cursors = [(0, 0), (3, 17), (6, 120), (11, 5), (20, 199)]
    q = np.zeros((len(cursors), 16)); v = np.zeros((len(cursors), 16))
    for n, (i_step, pos) in enumerate(cursors):
        set_refs_cursor(refs, i_step, pos)
        with contextlib.redirect_stdout(io.StringIO()):          # get_qvel prints a notice
            q[n], v[n] = np.asarray(refs.get_qpos(), float), np.asarray(refs.get_qvel(), float)
    np.savez_compressed(os.path.join(OUT, 'G13_hip3d.npz'), cursors=np.array(cursors, dtype=np.int32), q=q, v=v)
    print('wrote G13_hip3d.npz')


def make_q4_golden():
    """G15: quirk Q4 -- adjust_COM_Z_pos mutates the data set in place (base_ref_trajecs.py:126-127, called by reset_model, mimic_env.py:555-557).
    ONE environment lives through several episodes; its resets land on the same reference step at different positions, on a step an earlier
    episode rolled through, and (evaluation mode, quirk Q3) on step 0's table.  Driven: the reference's own reset_model() (the two random draws of
    get_random_init_state injected through the module's `random.randint`) and step() with injected dynamics; MuJoCo's set_state -> site_xpos is
    replaced by a numpy forward kinematics of the reference's MJCF (drloco_amd.mjcf._kinematics on parse_mjcf of walker3d_flat_feet.xml): what
    matters for Q4 is only that the foot height is translation-equivariant in the root's z, as it is in MuJoCo.  Recorded per reset: the draw, the
    lowest foot-site height the reference measured, the initial qpos, the COM-z row of the mutated step at a few samples; per control step: cursor,
    reference qpos (get_qpos: COM-x offset of Q1 and COM-z offsets of Q4 included), the three reward terms, the reward under a NON-default weight
    vector (com weight 0.3: where Q4 reaches the reward itself), Monitor.mean_ep_com_rew_smoothed of the reference's Monitor around the env."""
    refs_mod, walker_mod, monitor_mod, utils, hypers, MujocoException = _import_reference()
    import drloco.mujoco.mimic_env as mimic_env_mod
    import drloco.config.hypers as cfg
    mimic_env_mod.pause_mujoco_viewer_on_start = False
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from drloco_amd import mjcf
    model = mjcf.parse_mjcf(os.path.join(REF, 'drloco/mujoco/xml/walker3d_flat_feet.xml'), frame_skip=5)
    qpos_rows, qvel_rows = list(walker_mod.qpos_indices), list(walker_mod.qvel_indices)
    weights = [0.5, 0.2, 0.3, 0]
    saved_w = cfg.rew_weights
    cfg.rew_weights = weights
    env = make_env(walker_mod, refs_mod)
    utils._exp_weighted_averages.clear()

    def set_state(qpos, qvel):          # [3P] MujocoEnv.set_state -> mj_forward: only site_xpos is read afterwards (mimic_env.py:549)
        env.sim.data.qpos[:] = qpos; env.sim.data.qvel[:] = qvel
        xpos, xmat, _, _ = mjcf._kinematics(model, np.asarray(qpos, float))
        for s in range(model.nsite):
            b = model.site_body[s]
            env.sim.data.site_xpos[s] = xpos[b] + xmat[b] @ np.array(model.site_pos[s][:])
    env.set_state = set_state
    draws = []
    class _Rand:          # the module's `random`: randint(a, b) returns the injected draws in order (straight_walk_trajecs.py:466,471)
        @staticmethod
        def randint(a, b):
            x = draws.pop(0)
            assert a <= x <= b
            return x
    refs_mod.random = _Rand
    mon = monitor_mod.Monitor.__new__(monitor_mod.Monitor)
    mon.env = env
    mon.setup_containers = types.MethodType(monitor_mod.Monitor.setup_containers, mon)
    mon.num_dofs, mon.num_actions = 28, 8
    mon.setup_containers()
    r = np.random.default_rng(15)
    # (mode, i_step, pos, control steps before the injected fall)
    lens = [s.shape[1] for s in env.refs.data]
    episodes = [('train', 5, 40, 140), ('train', 5, 100, 130), ('train', 6, 10, 150), ('train', 5, lens[5] - 30, 60), ('train', 7, 3, 20),
                ('eval', -1, -1, 100), ('eval', -1, -1, 90), ('train', 0, 17, 40), ('train', 29, lens[29] - 8, 50)]
    T = sum(e[3] for e in episodes)
    out = dict(weights=np.array(weights[:3]), ep_mode=np.array([e[0] == 'eval' for e in episodes], np.int32), ep_draw=np.array([[e[1], e[2]] for e in episodes], np.int32),
               ep_len=np.array([e[3] for e in episodes], np.int32), ep_lowest=np.zeros(len(episodes)), ep_qpos0=np.zeros((len(episodes), 14)), ep_qvel0=np.zeros((len(episodes), 14)),
               ep_obs0=np.zeros((len(episodes), 29)), ep_cursor0=np.zeros((len(episodes), 3), np.int32), ep_zrow=np.zeros((len(episodes), 4)), ep_read_step=np.zeros(len(episodes), np.int32),
               actions=r.uniform(-1.3, 1.3, size=(T, 8)), inj_q=np.zeros((T, 14)), inj_v=np.zeros((T, 14)), ref_q=np.zeros((T, 14)), ref_v=np.zeros((T, 14)),
               i_step=np.zeros(T, np.int32), pos=np.zeros(T, np.int32), terms=np.zeros((T, 3)), rew=np.zeros(T), done=np.zeros(T, np.int32), obs=np.zeros((T, 29)),
               mean_ep_com_rew_smoothed=np.zeros(T), mean_ep_pos_rew_smoothed=np.zeros(T))
    state = {'q': None, 'v': None}
    def do_simulation(ctrl, n_frames):
        env.sim.data.qpos[:] = state['q']; env.sim.data.qvel[:] = state['v']
    env.do_simulation = do_simulation
    t = 0
    for e, (mode, i0, p0, n_steps) in enumerate(episodes):
        env._EVAL_MODEL = mode == 'eval'
        if mode == 'train':
            draws[:] = [i0, p0]
        site_before = None
        obs0 = env.reset_model()
        assert not draws
        # which table row did adjust_COM_Z_pos mutate: the one _qpos_full is bound to (data[i_step] after RSI, data[0] after an evaluation init: Q3)
        read_step = [k for k, d in enumerate(env.refs.data) if d is env.refs._qpos_full]
        assert len(read_step) == 1
        out['ep_read_step'][e] = read_step[0]
        out['ep_lowest'][e] = float(np.min(env.sim.data.site_xpos[:, 2]))          # after the second set_state: the residual height of the lowest site (~1e-17)
        out['ep_qpos0'][e], out['ep_qvel0'][e], out['ep_obs0'][e] = env.sim.data.qpos, env.sim.data.qvel, obs0
        out['ep_cursor0'][e] = env.refs._i_step, env.refs._pos, env.refs._trajec_len
        d = env.refs.data[read_step[0]]
        out['ep_zrow'][e] = [d[refs_mod.COM_POSZ, k] for k in (0, 1, d.shape[1] // 2, d.shape[1] - 1)]
        for k in range(n_steps):
            # injected end state of the control step: the reference sample the cursor will point at + noise (COM included, so that com_rew is not 1)
            shadow_pos = env.refs._pos + env.refs._increment
            if shadow_pos - env.refs._trajec_len + 1 > 0:
                q_ref = np.asarray(env.sim.data.qpos, float).copy(); v_ref = np.asarray(env.sim.data.qvel, float).copy()       # across a rollover: keep the last state
            else:
                save = env.refs._pos; env.refs._pos = shadow_pos
                q_ref, v_ref = np.asarray(env.refs.get_qpos(), float), np.asarray(env.refs.get_qvel(), float)
                env.refs._pos = save
            state['q'] = q_ref + 0.03 * r.standard_normal(14); state['v'] = v_ref + 0.3 * r.standard_normal(14)
            if k == n_steps - 1:
                state['q'][2] = 0.45          # the fall that ends the episode
            out['inj_q'][t], out['inj_v'][t] = state['q'], state['v']
            o, rew, done, _ = mon.step(out['actions'][t])
            out['ref_q'][t], out['ref_v'][t] = np.asarray(env.refs.get_qpos(), float), np.asarray(env.refs.get_qvel(), float)
            out['i_step'][t], out['pos'][t] = env.refs._i_step, env.refs._pos
            out['terms'][t] = env.pos_rew, env.vel_rew, env.com_rew
            out['rew'][t], out['done'][t], out['obs'][t] = rew, done, o
            out['mean_ep_com_rew_smoothed'][t], out['mean_ep_pos_rew_smoothed'][t] = mon.mean_ep_com_rew_smoothed, mon.mean_ep_pos_rew_smoothed
            assert bool(done) == (k == n_steps - 1)
            t += 1
    assert t == T
    # the history dependence is really in the fixture: the same step read with different COM-z offsets in different episodes
    assert len(set(np.round(out['ep_zrow'][[0, 1, 3], 0], 12).tolist())) == 3
    cfg.rew_weights = saved_w
    np.savez_compressed(os.path.join(OUT, 'G15_q4_com_z.npz'), **out)
    print('wrote G15_q4_com_z.npz')


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'hip3d':
        make_hip3d_golden()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'policy':
        make_policy_golden()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'ramp':
        make_ramp_golden()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'mocap':
        make_mocap_options_golden()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'q4':
        make_q4_golden()
        return
    refs_mod, walker_mod, monitor_mod, utils, hypers, MujocoException = _import_reference()
    import drloco.mujoco.mimic_env as mimic_env_mod
    mimic_env_mod.pause_mujoco_viewer_on_start = False
    rng = np.random.default_rng(20261001)
    qpos_rows = list(walker_mod.qpos_indices)
    qvel_rows = list(walker_mod.qvel_indices)

    # ---------------- G1: mocap table ----------------
    refs = refs_mod.StraightWalkingTrajectories(qpos_rows, qvel_rows)
    data = refs.data
    lens = np.array([s.shape[1] for s in data], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = qpos_rows + qvel_rows
    table = np.concatenate([np.asarray(s[rows, :], dtype=np.float64) for s in data], axis=1)
    np.savez_compressed(os.path.join(OUT, 'G1_mocap_table.npz'),
                        table=table, step_len=lens, step_off=offs,
                        qpos_rows=np.array(qpos_rows), qvel_rows=np.array(qvel_rows),
                        left_step_indices=np.array(refs.left_step_indices, dtype=np.int32),
                        step_velocities=np.asarray(refs.step_velocities, dtype=np.float64))

    # ---------------- G2: cursor traces ----------------
    starts = [(29, 270), (0, 0), (13, 100), (28, 10), (29, 0), (7, 254), (4, 248), (27, 278)]
    n_next = 3200
    tr = {k: np.zeros((len(starts), n_next), dtype=np.float64) for k in
          ['i_step', 'pos', 'len', 'dist', 'count_same_vel', 'phase', 'desvel', 'is_left', 'ref_ep_dur']}
    n_full = 400   # full 14+14 reference vectors for the first n_full calls, COM-x for all
    tr_q = np.zeros((len(starts), n_full, 14))
    tr_v = np.zeros((len(starts), n_full, 14))
    tr_comx = np.zeros((len(starts), n_next))
    count_in = np.zeros(len(starts), dtype=np.int32)
    for k, (i0, p0) in enumerate(starts):
        refs = refs_mod.StraightWalkingTrajectories(qpos_rows, qvel_rows)
        # exercise Q2: count_steps_same_vel persists; start some traces with a larger count
        refs.count_steps_same_vel = 1 + 3 * (k % 3)
        count_in[k] = refs.count_steps_same_vel
        set_refs_cursor(refs, i0, p0)
        for t in range(n_next):
            refs.next()
            tr['i_step'][k, t] = refs._i_step
            tr['pos'][k, t] = refs._pos
            tr['len'][k, t] = refs._trajec_len
            tr['dist'][k, t] = refs.dist
            tr['count_same_vel'][k, t] = refs.count_steps_same_vel
            tr['phase'][k, t] = refs.get_phase_variable()
            tr['desvel'][k, t] = refs.get_desired_walking_velocity_vector(False)[0]
            tr['is_left'][k, t] = refs.is_step_left()
            tr['ref_ep_dur'][k, t] = refs._ep_dur
            tr_comx[k, t] = float(refs.get_qpos()[0])
            if t < n_full:
                tr_q[k, t] = np.asarray(refs.get_qpos(), dtype=np.float64)
                tr_v[k, t] = np.asarray(refs.get_qvel(), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'G2_cursor_traces.npz'), starts=np.array(starts),
                        count_in=count_in, ref_qpos=tr_q, ref_qvel=tr_v, ref_comx=tr_comx, **tr)

    # ---------------- G3: reward / obs ----------------
    env = make_env(walker_mod, refs_mod)
    n = 256
    g3 = dict(i_step=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), count=np.zeros(n, np.int32),
              qpos=np.zeros((n, 14)), qvel=np.zeros((n, 14)),
              pose=np.zeros(n), vel=np.zeros(n), com=np.zeros(n), imit=np.zeros(n),
              obs=np.zeros((n, 29)), obs_nomirr=np.zeros((n, 29)), is_left=np.zeros(n, np.int32))
    for k in range(n):
        i0 = int(rng.integers(0, 30))
        p0 = int(rng.integers(0, lens[i0]))
        set_refs_cursor(env.refs, i0, p0)
        env.refs.count_steps_same_vel = int(rng.integers(1, 6))
        scale = [0.0, 0.01, 0.1, 0.5][k % 4]
        q = np.asarray(env.refs.get_qpos(), dtype=np.float64) + scale * rng.standard_normal(14)
        v = np.asarray(env.refs.get_qvel(), dtype=np.float64) + 10 * scale * rng.standard_normal(14)
        env.sim.data.qpos[:] = q
        env.sim.data.qvel[:] = v
        g3['i_step'][k], g3['pos'][k], g3['count'][k] = i0, p0, env.refs.count_steps_same_vel
        g3['qpos'][k], g3['qvel'][k] = q, v
        g3['imit'][k] = env.get_imitation_reward()
        g3['pose'][k], g3['vel'][k], g3['com'][k] = env.pos_rew, env.vel_rew, env.com_rew
        g3['obs'][k] = env._get_obs()
        g3['is_left'][k] = env.refs.is_step_left()
        # un-mirrored observation (MOD_MIRR_POLICY off)
        saved = hypers.modification
        hypers.modification = hypers.MOD_CUSTOM_POLICY
        g3['obs_nomirr'][k] = env._get_obs()
        hypers.modification = saved
    np.savez_compressed(os.path.join(OUT, 'G3_reward_obs.npz'), **g3)

    # ---------------- G5: actions ----------------
    acts = np.concatenate([rng.uniform(-2, 2, size=(60, 8)),
                           np.array([[0.1, -0.2, -0.3, 0.4, 0.5, 0.2, 1.0, -1.0]]),
                           np.zeros((1, 8)), np.array([[1.5, -1.5, 1, -1, 0.999, -0.999, 1e-9, -1e-9]])])
    resc = np.array([env._rescale_actions(a) for a in acts])
    mirr = np.array([env.mirror_action(r.copy()) for r in resc])
    np.savez_compressed(os.path.join(OUT, 'G5_actions.npz'), actions=acts, rescaled=resc, mirrored=mirr)

    # ---------------- G4: step() traces with injected dynamics ----------------
    # The "dynamics" are injected: do_simulation copies the next row of a prepared
    # (qpos, qvel) stream into sim.data; an exception can be raised at a chosen step.
    def run_trace(i0, p0, count0, T, fall_at=None, exc_at=None, ep_dur0=0, seed=0, rsi_after_exc=(3, 17)):
        r = np.random.default_rng(seed)
        env = make_env(walker_mod, refs_mod)
        set_refs_cursor(env.refs, i0, p0)
        env.refs.count_steps_same_vel = count0
        env.ep_dur = ep_dur0
        # state stream: reference + noise (so that rewards are non-trivial)
        stream_q = np.zeros((T, 14))
        stream_v = np.zeros((T, 14))
        shadow = refs_mod.StraightWalkingTrajectories(qpos_rows, qvel_rows)
        set_refs_cursor(shadow, i0, p0)
        for t in range(T):
            shadow.next()
            stream_q[t] = np.asarray(shadow.get_qpos(), float) + 0.05 * r.standard_normal(14)
            stream_v[t] = np.asarray(shadow.get_qvel(), float) + 0.5 * r.standard_normal(14)
            if fall_at is not None and t == fall_at:
                stream_q[t, 2] = 0.49
        # large velocities to exercise the 5.5 clip in update_walked_distance
        stream_v[T // 2, 0] = 7.0
        stream_v[T // 2, 1] = -6.0
        state = {'t': 0}
        ctrls = []

        def do_simulation(ctrl, n_frames):
            ctrls.append(np.array(ctrl, dtype=np.float64))
            if exc_at is not None and state['t'] == exc_at:
                raise MujocoException('injected')
            env.sim.data.qpos[:] = stream_q[state['t']]
            env.sim.data.qvel[:] = stream_v[state['t']]
            state['t'] += 1
        env.do_simulation = do_simulation

        def reset():
            # stands in for MujocoEnv.reset() -> reset_model() with injected RSI draw and
            # FK replaced by "no shift" (FK is MuJoCo's; covered by the build's own tests)
            env.ep_dur = 0
            env.walked_distance = 0
            set_refs_cursor(env.refs, *rsi_after_exc)
            env.sim.data.qpos[:] = np.asarray(env.refs.get_qpos(), float)
            env.sim.data.qvel[:] = np.asarray(env.refs.get_qvel(), float)
            env.refs.next()
            return env._get_obs()
        env.reset = reset
        actions = r.uniform(-1.3, 1.3, size=(T, 8))
        out = dict(actions=actions, stream_q=stream_q, stream_v=stream_v,
                   obs=np.zeros((T, 29)), rew=np.zeros(T), done=np.zeros(T, np.int32),
                   walked=np.zeros(T), ep_dur=np.zeros(T, np.int32), ctrl=np.zeros((T, 8)),
                   pos_rew=np.zeros(T), vel_rew=np.zeros(T), com_rew=np.zeros(T),
                   i_step=np.zeros(T, np.int32), pos=np.zeros(T, np.int32), nsteps=0)
        for t in range(T):
            o, rew, done, info = env.step(actions[t])
            out['obs'][t], out['rew'][t], out['done'][t] = o, rew, done
            out['walked'][t], out['ep_dur'][t] = env.walked_distance, env.ep_dur
            out['ctrl'][t] = ctrls[-1]
            out['pos_rew'][t], out['vel_rew'][t], out['com_rew'][t] = env.pos_rew, env.vel_rew, env.com_rew
            out['i_step'][t], out['pos'][t] = env.refs._i_step, env.refs._pos
            out['nsteps'] = t + 1
            if done:
                break
        out['start'] = np.array([i0, p0, count0, ep_dur0])
        out['rsi_after_exc'] = np.array(rsi_after_exc)
        out['rew_signbit'] = np.signbit(out['rew']).astype(np.int32)
        return out

    g4 = {}
    cases = dict(fall=run_trace(5, 40, 1, 60, fall_at=37, seed=1),
                 timeout=run_trace(28, 200, 4, 80, ep_dur0=2950, seed=2),
                 exception=run_trace(12, 3, 2, 40, exc_at=21, seed=3),
                 rollover=run_trace(29, 250, 1, 300, seed=4))
    for cname, c in cases.items():
        for k, v in c.items():
            g4[f'{cname}__{k}'] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, 'G4_step_traces.npz'), **g4)

    # ---------------- G6: do_terminate_early truth table ----------------
    env = make_env(walker_mod, refs_mod)
    n = 128
    g6 = dict(i_step=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), qpos=np.zeros((n, 14)),
              flags=np.zeros((n, 4), np.int32))
    for k in range(n):
        i0 = int(rng.integers(0, 30))
        p0 = int(rng.integers(0, lens[i0]))
        set_refs_cursor(env.refs, i0, p0)
        q = np.asarray(env.refs.get_qpos(), dtype=np.float64)
        mode = k % 8
        if mode == 1: q[2] = 0.74
        if mode == 2: q[1] = 0.21 * (1 if k % 16 < 8 else -1)
        if mode == 3: q[4] = 0.31
        if mode == 4: q[4] = -0.06
        if mode == 5: q[3] += 0.21
        if mode == 6: q[5] += 0.6      # axial deviation is ignored
        if mode == 7: q += 0.05 * rng.standard_normal(14)
        env.sim.data.qpos[:] = q
        g6['i_step'][k], g6['pos'][k], g6['qpos'][k] = i0, p0, q
        g6['flags'][k] = [int(bool(x)) for x in env.do_terminate_early()]
    np.savez_compressed(os.path.join(OUT, 'G6_terminate_early.npz'), **g6)

    # ---------------- G7: Monitor smoothing traces ----------------
    utils._exp_weighted_averages.clear()

    class FakeEnv:
        """Feeds prepared (reward, done, components, torque) streams into Monitor.step."""
        def __init__(self, T, seed):
            r = np.random.default_rng(seed)
            self.rew = r.uniform(0.2, 1.2, T)
            self.done = np.zeros(T, bool)
            t = 0
            while True:
                t += int(r.integers(3, 40))
                if t >= T:
                    break
                self.done[t] = True
            self.rew[self.done] = 0.0
            self.comp = r.uniform(0, 1, (T, 3))
            self.tor = r.uniform(0, 300, T)
            self.cursor = r.integers(0, 270, T)
            self.walked = np.cumsum(r.uniform(0, 0.01, T))
            self.t = -1
            self.refs = types.SimpleNamespace(_pos=0, get_kinematics_labels=lambda: ['x'] * 28)
            self.action_space = types.SimpleNamespace(high=np.zeros(8))
            self.pos_rew = self.vel_rew = self.com_rew = 0

        def step(self, a):
            self.t += 1
            t = self.t
            self.refs._pos = int(self.cursor[t])
            self.pos_rew, self.vel_rew, self.com_rew = self.comp[t]
            return None, self.rew[t], bool(self.done[t]), {}

        def get_actuator_torques(self, abs_mean=False):
            return self.tor[self.t]

        def get_walked_distance(self):
            return self.walked[self.t]

    T = 600
    fe = FakeEnv(T, 77)
    mon = monitor_mod.Monitor.__new__(monitor_mod.Monitor)
    mon.env = fe
    mon.setup_containers = types.MethodType(monitor_mod.Monitor.setup_containers, mon)
    mon.num_dofs, mon.num_actions = 28, 8
    mon.setup_containers()
    names = ['ep_len_smoothed', 'ep_ret_smoothed', 'mean_reward_smoothed', 'moved_distance',
             'mean_ep_pos_rew_smoothed', 'mean_ep_vel_rew_smoothed', 'mean_ep_com_rew_smoothed',
             'mean_abs_ep_torque_smoothed', 'median_abs_torque_smoothed']
    g7 = {k: np.zeros(T) for k in names}
    for t in range(T):
        mon.step(None)
        for k in names:
            g7[k][t] = getattr(mon, k)
    g7.update(rew=fe.rew, done=fe.done.astype(np.int32), comp=fe.comp, tor=fe.tor, cursor=fe.cursor,
              walked=fe.walked, ep_lens=np.array(mon.ep_lens), rsi_positions=np.array(mon.rsi_positions),
              et_positions=np.array(mon.et_positions),
              difficult_rsi_phases=np.array(mon.difficult_rsi_phases))
    np.savez_compressed(os.path.join(OUT, 'G7_monitor.npz'), **g7)
    # ---------------- G8: deterministic (evaluation) init states ----------------
    env = make_env(walker_mod, refs_mod)
    env._EVAL_MODEL = True
    n = 45
    g8 = dict(i_step=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), trajec_len=np.zeros(n, np.int32),
              qpos=np.zeros((n, 14)), qvel=np.zeros((n, 14)), is_left=np.zeros(n, np.int32))
    for k in range(n):
        q, v = env.get_init_state(not env.is_evaluation_on() and not env._FOLLOW_DESIRED_SPEED_PROFILE)
        g8['i_step'][k], g8['pos'][k], g8['trajec_len'][k] = env.refs._i_step, env.refs._pos, env.refs._trajec_len
        g8['qpos'][k], g8['qvel'][k] = np.asarray(q, float), np.asarray(v, float)
        g8['is_left'][k] = env.refs.is_step_left()
    # cursor trace after one evaluation init (k = 0 again after 45 = 2*20 + 5 -> next is k = 5)
    q, v = env.get_init_state(False)
    T = 160
    tr = dict(t_i_step=np.zeros(T, np.int32), t_pos=np.zeros(T, np.int32), t_len=np.zeros(T, np.int32), t_comx=np.zeros(T), t_phase=np.zeros(T))
    tr['t_start'] = np.array([env.refs._i_step, env.refs._pos, env.refs._trajec_len])
    for t in range(T):
        env.refs.next()
        tr['t_i_step'][t], tr['t_pos'][t], tr['t_len'][t] = env.refs._i_step, env.refs._pos, env.refs._trajec_len
        tr['t_comx'][t] = float(env.refs.get_qpos()[0])
        tr['t_phase'][t] = env.refs.get_phase_variable()
    np.savez_compressed(os.path.join(OUT, 'G8_eval_init.npz'), **g8, **tr)
    make_loco3d_golden(mimic_env_mod, hypers, rng)
    print('golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
