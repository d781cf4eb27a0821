"""The oracle's environment logic against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Bit-exact unless noted."""
import os
import sys

import numpy as np
import pytest

from drloco_amd import abi
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:      # NpzFile re-reads on every access
        return {k: z[k] for k in z.files}


def cursor(i_step, pos, count=1, ep_dur=0, n=1):
    c = np.zeros((abi.DL_CUR_WORDS, n), np.int32)
    c[abi.DL_CUR_I_STEP] = i_step
    c[abi.DL_CUR_POS] = pos
    c[abi.DL_CUR_RSI_STEP] = i_step
    c[abi.DL_CUR_READ_STEP] = i_step
    c[abi.DL_CUR_COUNT] = count
    c[abi.DL_CUR_EP_DUR] = ep_dur
    return c


def make_env(oracle, model, refs, n=1, **kw):
    return oracle.OracleEnv(model, refs, abi.default_config(**kw), n)


def test_G1_table(refs):
    g = load('G1_mocap_table.npz')
    assert np.array_equal(refs.table, g['table'])
    assert np.array_equal(refs.step_off, g['step_off'])
    assert np.array_equal(np.nonzero(refs.step_is_left)[0], g['left_step_indices'])
    assert np.array_equal(refs.step_vel, g['step_velocities'])
    assert refs.stride == 2 and refs.n_steps == 30 and refs.table.shape == (28, 7906)


def test_G2_cursor_traces(oracle, model, refs):
    g = load('G2_cursor_traces.npz')
    K, T = g['i_step'].shape
    env = make_env(oracle, model, refs, n=K, ep_dur_max=10 ** 9)
    starts = g['starts']
    cur = cursor(starts[:, 0], starts[:, 1], g['count_in'], n=K)
    env.set_state(cursor=cur)
    q_up = np.array(model.jnt_qpos0[:14])
    for t in range(T):
        for k in range(K):
            env.inject_state(k, q_up, np.zeros(14))
        obs, rew, done, _, _ = env.step(np.zeros((K, 8)))
        assert not done.any()
        st = env.get_state()['cursor']
        assert np.array_equal(st[abi.DL_CUR_I_STEP], g['i_step'][:, t].astype(int)), t
        assert np.array_equal(st[abi.DL_CUR_POS], g['pos'][:, t].astype(int)), t
        assert np.array_equal(st[abi.DL_CUR_COUNT], g['count_same_vel'][:, t].astype(int)), t
        assert np.array_equal(obs[:, 0], g['phase'][:, t]), t
        assert np.array_equal(obs[:, 1], g['desvel'][:, t]), t
        for k in range(K):
            qr, vr = env.ref_lookup(k)
            assert qr[0] == g['ref_comx'][k, t], (k, t)
            if t < g['ref_qpos'].shape[1]:
                assert np.array_equal(qr, g['ref_qpos'][k, t]) and np.array_equal(vr, g['ref_qvel'][k, t])
        lens = refs.step_len[st[abi.DL_CUR_READ_STEP]]
        assert np.array_equal(lens, g['len'][:, t].astype(int))
        assert np.array_equal(refs.step_is_left[st[abi.DL_CUR_I_STEP]], g['is_left'][:, t].astype(int))


def test_G3_reward_obs(oracle, model, refs):
    g = load('G3_reward_obs.npz')
    n = len(g['i_step'])
    for mirror, key in ((1, 'obs'), (0, 'obs_nomirr')):
        env = make_env(oracle, model, refs, n=n, mirror_policy=mirror, rew_weights=(0.8, 0.2, 0.0))
        env.set_state(qpos=g['qpos'].T, qvel=g['qvel'].T, cursor=cursor(g['i_step'], g['pos'], g['count'], n=n))
        obs, imit, terms = env.observe()
        assert np.array_equal(obs, g[key])
        np.testing.assert_allclose(terms[:, 0], g['pose'], rtol=5e-16)
        np.testing.assert_allclose(terms[:, 1], g['vel'], rtol=5e-16)
        np.testing.assert_allclose(terms[:, 2], g['com'], rtol=5e-16)
        np.testing.assert_allclose(imit, g['imit'], rtol=5e-16)
    assert g['is_left'].sum() > 20 and (1 - g['is_left']).sum() > 20


def test_G5_actions(oracle, model, refs):
    g = load('G5_actions.npz')
    n = len(g['actions'])
    env = make_env(oracle, model, refs, n=2 * n, ep_dur_max=10 ** 9)
    # first half on a right step (even index), second half on a left step (odd index)
    istep = np.concatenate([np.full(n, 4), np.full(n, 5)])
    env.set_state(cursor=cursor(istep, 10, n=2 * n))
    q_up = np.array(model.jnt_qpos0[:14])
    for k in range(2 * n):
        env.inject_state(k, q_up, np.zeros(14))
    env.step(np.concatenate([g['actions'], g['actions']]))
    ctrl = env.last_ctrl()
    assert np.array_equal(ctrl[:n], g['rescaled'])
    assert np.array_equal(ctrl[n:], g['mirrored'])
    # an action of exactly 0 maps to |0|*low = -0.0 (mimic_env.py:190)
    z = np.where((g['actions'] == 0).all(axis=1))[0][0]
    assert np.signbit(ctrl[z]).all()


@pytest.mark.parametrize('case', ['fall', 'timeout', 'exception', 'rollover'])
def test_G4_step_traces(oracle, model, refs, case):
    g = load('G4_step_traces.npz')
    G = {k.split('__')[1]: g[k] for k in g if k.startswith(case + '__')}
    i0, p0, count0, ep0 = G['start']
    T = int(G['nsteps'])
    env = make_env(oracle, model, refs, n=1)
    env.set_state(cursor=cursor(i0, p0, count0, ep0))
    if case == 'exception':
        env.inject_rsi(0, *G['rsi_after_exc'])
    t_state = 0
    for t in range(T):
        if case == 'exception' and t == T - 1:
            env.inject_exception(0)
        else:
            env.inject_state(0, G['stream_q'][t_state], G['stream_v'][t_state])
            t_state += 1
        obs, rew, done, term, terms = env.step(G['actions'][t][None])
        assert np.array_equal(env.last_ctrl()[0], G['ctrl'][t])
        assert int(done[0]) == int(G['done'][t]), t
        assert abs(rew[0] - G['rew'][t]) <= 5e-16 * abs(G['rew'][t]) and np.signbit(rew[0]) == bool(G['rew_signbit'][t]), t
        if not done[0]:
            assert np.array_equal(obs[0], G['obs'][t]), t
            st = env.get_state()
            assert st['walked'][0] == G['walked'][t]
            assert st['cursor'][abi.DL_CUR_EP_DUR, 0] == G['ep_dur'][t]
            assert st['cursor'][abi.DL_CUR_I_STEP, 0] == G['i_step'][t] and st['cursor'][abi.DL_CUR_POS, 0] == G['pos'][t]
            np.testing.assert_allclose(terms[0], [G['pos_rew'][t], G['vel_rew'][t], G['com_rew'][t]], rtol=5e-16)
        elif case == 'exception':
            # step() returned reset()'s observation; the oracle's reset additionally puts the lowest
            # foot corner on the floor (FK, not part of the stubbed reference run): skip COM-z (obs[3])
            keep = np.arange(29) != 3
            assert np.array_equal(term[0][keep], G['obs'][t][keep])
            assert np.array_equal(obs[0][keep], G['obs'][t][keep])   # second reset, same injected draw
            assert env.get_state()['cursor'][abi.DL_CUR_EPISODE, 0] == 2
        else:
            assert np.array_equal(term[0], G['obs'][t]), t
    assert done[0] == (0 if case == 'rollover' else 1)
    if case == 'fall':
        assert np.signbit(rew[0])          # -1 * 0 = -0.0
    if case == 'timeout':
        assert not np.signbit(rew[0]) and G['ep_dur'][T - 1] == 3000


def test_G6_terminate_early(oracle, model, refs):
    g = load('G6_terminate_early.npz')
    n = len(g['i_step'])
    env = make_env(oracle, model, refs, n=n)
    env.set_state(qpos=g['qpos'].T, cursor=cursor(g['i_step'], g['pos'], n=n))
    for k in range(n):
        assert np.array_equal(env.terminate_early(k), g['flags'][k]), k
    assert g['flags'][:, 0].sum() > 10


def test_G7_monitor(oracle, model, refs):
    g = load('G7_monitor.npz')
    env = make_env(oracle, model, refs, n=1)
    names = ['ep_len_smoothed', 'ep_ret_smoothed', 'mean_reward_smoothed', 'moved_distance',
             'mean_ep_pos_rew_smoothed', 'mean_ep_vel_rew_smoothed', 'mean_ep_com_rew_smoothed',
             'mean_abs_ep_torque_smoothed']
    for t in range(len(g['rew'])):
        env.monitor_feed(0, g['rew'][t], g['done'][t], *g['comp'][t], g['tor'][t], g['walked'][t])
        for k in names:
            np.testing.assert_allclose(env.stats(k)[0], g[k][t], rtol=1e-12, atol=1e-300, err_msg=f'{k} t={t}')


def test_reset_places_lowest_foot_corner_on_floor(oracle, model, refs):
    rng = np.random.default_rng(0)
    n = 64
    env = make_env(oracle, model, refs, n=n)
    steps = rng.integers(0, 30, n).astype(np.int32)
    pos = (rng.random(n) * refs.step_len[steps]).astype(np.int32)
    obs = env.reset(init_step=steps, init_pos=pos)
    st = env.get_state()
    for k in range(n):
        r = oracle.probe_forward(model, st['qpos'][:, k], st['qvel'][:, k])
        assert abs(r['site_xpos'][:, 2].min()) < 1e-12
        base = refs.step_off[steps[k]] + pos[k]
        want = refs.table[:14, base].copy()
        assert np.array_equal(np.delete(st['qpos'][:, k], 2), np.delete(want, 2))
        assert np.array_equal(st['qvel'][:, k], refs.table[14:, base])
    # cursor advanced once (mimic_env.py:568)
    assert np.array_equal(st['cursor'][abi.DL_CUR_EP_DUR], np.zeros(n))
    assert (st['cursor'][abi.DL_CUR_EPISODE] == 1).all()
    # imitation reward right after RSI is 1 up to the one-sample cursor advance (mimic_env.py:562-564)
    _, imit, _ = env.observe()
    assert (imit > 0.8).all()


def test_rsi_stream_is_deterministic_and_in_range(oracle, refs):
    seen = set()
    for env_id in range(200):
        for ep in range(5):
            s, p = oracle.rsi_draw(1234, env_id, ep, refs.step_off)
            assert 0 <= s < 30 and 0 <= p < refs.step_len[s]
            assert (s, p) == oracle.rsi_draw(1234, env_id, ep, refs.step_off)
            seen.add(s)
    assert len(seen) == 30


def test_G8_evaluation_init(oracle, model, refs):
    """_get_deterministic_init_state incl. quirk Q3 (state read from step 0's table)."""
    g = load('G8_eval_init.npz')
    env = make_env(oracle, model, refs, n=1, ep_dur_max=10 ** 9)
    env.set_eval(True)
    n = len(g['i_step'])
    for k in range(n):
        env.reset()
        st = env.get_state()
        cur = st['cursor'][:, 0]
        assert cur[abi.DL_CUR_I_STEP] == g['i_step'][k] == k % 20
        assert cur[abi.DL_CUR_POS] - 2 == g['pos'][k]          # reset_model advances the cursor once
        assert cur[abi.DL_CUR_READ_STEP] == 0 and refs.step_len[0] == g['trajec_len'][k]
        assert cur[abi.DL_CUR_EVAL_K] == (k + 1) % 20
        assert np.array_equal(np.delete(st['qpos'][:, 0], 2), np.delete(g['qpos'][k], 2))
        assert np.array_equal(st['qvel'][:, 0], g['qvel'][k])
    # cursor trace after the next evaluation init: rolls from step 0's table into step k+1 with step k's end-x
    env.reset()
    q_up = np.array(model.jnt_qpos0[:14])
    T = len(g['t_i_step'])
    for t in range(T):
        st = env.get_state()['cursor'][:, 0]
        assert st[abi.DL_CUR_I_STEP] == g['t_i_step'][t] and st[abi.DL_CUR_POS] == g['t_pos'][t], t
        assert refs.step_len[st[abi.DL_CUR_READ_STEP]] == g['t_len'][t]
        qr, _ = env.ref_lookup(0)
        assert qr[0] == g['t_comx'][t], t
        env.inject_state(0, q_up, np.zeros(14))
        obs, _, done, _, _ = env.step(np.zeros((1, 8)))
        assert not done[0]
    assert g['t_i_step'].max() > g['t_start'][0]      # the trace crossed a rollover


def _loco3d_setup(oracle, n, g):
    from drloco_amd import mocap, models
    ang, vel = mocap.synthetic_loco3d(L=int(g['L']), seed=int(g['seed']))
    np.testing.assert_allclose([ang.sum(), vel.sum(), np.abs(ang).sum()], g['table_checksum'], rtol=1e-13)
    assert list(g['qpos_rows']) == mocap.LOCO3D_ROWS and int(g['stride']) == 5
    table = mocap.loco3d_table(ang, vel)
    model = models.make_model(models.WALKER_165CM)
    env = oracle.OracleEnv(model, table, abi.loco3d_config(ep_dur_max=10 ** 9), n)
    return env, model, table


def test_G9_loco3d_cursor_and_lookup(oracle):
    """Loco3dReferenceTrajectories: base-class cursor (wrap to 0), lookups, desired velocity."""
    g = load('G9_loco3d.npz')
    env, model, table = _loco3d_setup(oracle, 1, g)
    cur = cursor(0, int(g['c_start']))
    env.set_state(cursor=cur)
    q_up = np.array(model.jnt_qpos0[:19])
    for t in range(len(g['c_pos'])):
        env.inject_state(0, q_up, np.zeros(19))
        obs, _, done, _, _ = env.step(np.zeros((1, 13)))
        assert env.get_state()['cursor'][abi.DL_CUR_POS, 0] == g['c_pos'][t], t
        qr, vr = env.ref_lookup(0)
        assert np.array_equal(qr, g['c_q'][t]) and np.array_equal(vr, g['c_v'][t])
        np.testing.assert_allclose(obs[0, 8:10], g['c_desvel'][t], rtol=1e-14, atol=1e-16)
    assert (np.diff(g['c_pos']) < 0).any()          # the trace wrapped


def test_G9_loco3d_obs_and_reward(oracle):
    g = load('G9_loco3d.npz')
    n = len(g['o_pos'])
    env, model, table = _loco3d_setup(oracle, n, g)
    env.set_state(qpos=g['o_q'].T, qvel=g['o_v'].T, cursor=cursor(0, g['o_pos'], n=n))
    obs, imit, terms = env.observe()
    assert obs.shape == (n, 47)
    assert np.isnan(g['o_obs'][0, 8:10]).all() and np.isnan(obs[0, 8:10]).all()      # cursor on the last sample: empty mean
    np.testing.assert_allclose(obs[1:], g['o_obs'][1:], rtol=2e-15, atol=1e-16)
    np.testing.assert_allclose(terms, g['o_terms'], rtol=5e-16)
    np.testing.assert_allclose(imit, g['o_imit'], rtol=5e-16)


def test_G9_loco3d_step_trace(oracle):
    g = load('G9_loco3d.npz')
    env, model, table = _loco3d_setup(oracle, 1, g)
    env.set_state(cursor=cursor(0, int(g['s_start'])))
    for t in range(len(g['s_rew'])):
        env.inject_state(0, g['s_q'][t], g['s_v'][t])
        obs, rew, done, _, _ = env.step(g['s_actions'][t][None])
        assert np.array_equal(env.last_ctrl()[0], g['s_ctrl'][t])          # 13 motors, no mirroring
        assert int(done[0]) == int(g['s_done'][t])
        np.testing.assert_allclose(obs[0], g['s_obs'][t], rtol=2e-15, atol=1e-16)
        assert abs(rew[0] - g['s_rew'][t]) <= 5e-16 * abs(g['s_rew'][t])
        st = env.get_state()
        assert st['cursor'][abi.DL_CUR_POS, 0] == g['s_pos'][t] and st['walked'][0] == g['s_walked'][t]


def test_G10_policy_trunk(oracle):
    """The numpy restatement of the policy forward reproduces the reference's CustomHiddenLayers (shared trunk)."""
    with np.load(os.path.join(GOLDEN, 'G10_policy_trunk.npz')) as z:
        g = {k: z[k] for k in z.files}
    assert int(g['shared']) == 1 and int(g['latent_dim_pi']) == 64
    A = 8
    wa = np.zeros((A, 64)); wv = np.zeros((1, 64))
    lat, act, val, logp = oracle.policy_forward(g['w1'], g['b1'], g['w2'], g['b2'], wa, np.zeros(A), wv, np.zeros(1), np.full(A, -0.75),
                                                g['x'], np.zeros((len(g['x']), A)))
    np.testing.assert_allclose(lat, g['latent'], atol=2e-6)          # torch float32 vs numpy float64
    assert np.allclose(logp, A * (0.75 - 0.5 * np.log(2 * np.pi)))


def test_G11_mirrored_refs(oracle, model, refs):
    """StraightWalkingTrajectories(mirror_refs=True): the table RefTable.mirrored() builds, and the oracle's cursor /
    lookup driven on it across a right -> mirrored-left rollover."""
    g = load('G11_mocap_options.npz')
    m = refs.mirrored()
    assert np.array_equal(m.step_len, g['m_step_len'])
    assert np.array_equal(np.nonzero(m.step_is_left)[0], g['m_left']) and np.array_equal(m.step_vel, g['m_step_vel'])
    assert np.array_equal(m.table[:, :m.step_off[6]], g['m_table6'])
    for i in range(m.n_steps):
        blk = m.table[:, m.step_off[i]:m.step_off[i + 1]]
        np.testing.assert_allclose(blk.sum(axis=1), g['m_rowsum'][i], rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(np.abs(blk).sum(axis=1), g['m_rowabs'][i], rtol=1e-13, atol=1e-12)
    # left steps really are mirrored right steps: the lateral COM position flips sign
    assert np.array_equal(m.table[1, m.step_off[1]:m.step_off[2]], -m.table[1, m.step_off[0]:m.step_off[1]])
    env = make_env(oracle, model, m, n=1, ep_dur_max=10 ** 9)
    env.set_state(cursor=cursor(int(g['m_start'][0]), int(g['m_start'][1])))
    q_up = np.array(model.jnt_qpos0[:14])
    for t in range(len(g['m_q'])):
        env.inject_state(0, q_up, np.zeros(14))
        env.step(np.zeros((1, 8)))
        st = env.get_state()['cursor']
        assert (st[abi.DL_CUR_I_STEP, 0], st[abi.DL_CUR_POS, 0]) == tuple(g['m_cur'][t]), t
        qr, vr = env.ref_lookup(0)
        assert np.array_equal(qr, g['m_q'][t]) and np.array_equal(vr, g['m_v'][t]), t
    assert len(set(g['m_cur'][:, 0])) > 1


def test_G11_adapt_trajectories():
    """BaseReferenceTrajectories.adapt_trajectories through Loco3dReferenceTrajectories (synthetic table, reference schema)."""
    from drloco_amd import mocap
    g = load('G11_mocap_options.npz')
    ang, vel = mocap.synthetic_loco3d(L=int(g['a_L']), seed=int(g['a_seed']))
    t = mocap.loco3d_table(ang, vel, adaptations=dict(zip(g['a_rows'].tolist(), g['a_scalars'].tolist())))
    assert np.array_equal(t.table[:19], g['a_q']) and np.array_equal(t.table[19:], g['a_v'])
    plain = mocap.loco3d_table(ang, vel)
    assert not np.array_equal(plain.table, t.table)


def test_G14_ramp_layout_table(ramp_refs):
    """convert_straight_walk_mat on the 40-row layout == what the UNPATCHED reference module holds after loading the same file: table rows
    (trunk Euler angles from rows 37-39, not the GRF rows), step lengths, left-step list, smoothed step velocities -- bit for bit."""
    table, g = ramp_refs
    assert table.n_steps == 250 and table.table.shape[0] == 28
    assert np.array_equal(table.step_len, g['step_len'])
    assert np.array_equal(np.nonzero(table.step_is_left)[0], g['left_step_indices'])
    assert np.array_equal(table.step_vel, g['step_velocities'])
    assert list(g['qpos_rows'][3:6]) == [37, 38, 39]
    o = table.step_off
    assert np.array_equal(table.table[:, :o[3]], g['table_head']) and np.array_equal(table.table[:, o[-3]:], g['table_tail'])
    for i in range(table.n_steps):
        blk = table.table[:, o[i]:o[i + 1]]
        assert np.array_equal(blk.sum(axis=1), g['rowsum'][i]) and np.array_equal(np.abs(blk).sum(axis=1), g['rowabs'][i]), i
    # the 38-row layout keeps working and a foreign layout is refused
    from drloco_amd import mocap
    import scipy.io as spio
    with pytest.raises(ValueError, match='rows per step'):
        p = os.path.join(os.path.dirname(GOLDEN), '_bad.mat')
        try:
            d = np.empty((1, 2), dtype=object); d[0, 0] = np.zeros((39, 10)); d[0, 1] = np.zeros((39, 10))
            spio.savemat(p, {'Data': d})
            mocap.convert_straight_walk_mat(p)
        finally:
            if os.path.exists(p):
                os.remove(p)


def test_G14_ramp_layout_cursor(oracle, model, ramp_refs):
    """The oracle's cursor on the 250-step ramp table against the reference's own next() traces: step index, position, counter (quirk Q2),
    phase, desired velocity (moving through changing step speeds and across the wrap after the last step), reference sample."""
    table, g = ramp_refs
    starts = g['starts']
    K, T = g['i_step'].shape
    env = make_env(oracle, model, table, n=K, ep_dur_max=10 ** 9)
    env.set_state(cursor=cursor(starts[:, 0], starts[:, 1], starts[:, 2], n=K))
    q_up = np.array(model.jnt_qpos0[:14])
    for t in range(T):
        for k in range(K):
            env.inject_state(k, q_up, np.zeros(14))
        obs, rew, done, _, _ = env.step(np.zeros((K, 8)))
        assert not done.any()
        st = env.get_state()['cursor']
        assert np.array_equal(st[abi.DL_CUR_I_STEP], g['i_step'][:, t].astype(int)), t
        assert np.array_equal(st[abi.DL_CUR_POS], g['pos'][:, t].astype(int)), t
        assert np.array_equal(st[abi.DL_CUR_COUNT], g['count_same_vel'][:, t].astype(int)), t
        assert np.array_equal(obs[:, 0], g['phase'][:, t]) and np.array_equal(obs[:, 1], g['desvel'][:, t]), t
        assert np.array_equal(table.step_len[st[abi.DL_CUR_READ_STEP]], g['len'][:, t].astype(int))
        assert np.array_equal(table.step_is_left[st[abi.DL_CUR_I_STEP]], g['is_left'][:, t].astype(int))
        for k in range(K):
            qr, vr = env.ref_lookup(k)
            assert np.array_equal(qr, g['ref_qpos'][k, t]) and np.array_equal(vr, g['ref_qvel'][k, t]), (k, t)
    assert len(set(g['desvel'].reshape(-1).tolist())) >= 5 and (g['i_step'][2] == 0).any() | (g['i_step'][2] == 1).any()


def test_G13_hip3d_rows(oracle, model, refs):
    """StraightWalking3dHipTrajectories (straight_walk_hip3d_trajecs.py:8-19): the 16-value rows the class returns at a few cursors,
    from the oracle's reference lookup on the packaged table + mocap.hip3d_qpos / hip3d_qvel."""
    from drloco_amd import mocap
    g = load('G13_hip3d.npz')
    env = make_env(oracle, model, refs, n=1, ep_dur_max=10 ** 9)
    for n, (i_step, pos) in enumerate(g['cursors']):
        env.set_state(cursor=cursor(int(i_step), int(pos)))
        qr, vr = env.ref_lookup(0)
        assert np.array_equal(mocap.hip3d_qpos(qr), g['q'][n]) and np.array_equal(mocap.hip3d_qvel(vr), g['v'][n]), n
    assert mocap.hip3d_qpos(np.zeros((3, 14))).shape == (3, 16)


# ---------------------------------------------------------------------------------------------
# G12: vectors of the REAL MuJoCo (tools/dump_mujoco_vectors.py).  The file can only be produced where `import mujoco` or
# `import mujoco_py` works -- not in the build container -- so these tests skip LOUDLY until somebody commits it; with it the
# dynamics part of the oracle is pinned to the binary and DESIGN.md 2's "parity unpinned" goes away.
G12 = os.path.join(GOLDEN, 'G12_mujoco_step.npz')
G12_SKIP = ('tests/golden/G12_mujoco_step.npz is absent: the dynamics oracle is NOT pinned to a real MuJoCo build (PARITY UNPINNED, DESIGN.md 2). '
            'Run tools/dump_mujoco_vectors.py on a machine where `import mujoco` or `import mujoco_py` works and commit the file.')


def g12_models():
    from drloco_amd import models
    return {'straight': models.make_model(), 'walker165': models.make_model(models.WALKER_165CM)}


def check_g12(g, oracle, key):
    m = g12_models()[key]
    nv = m.nv
    # mj_setConst quantities the constraint regularisation depends on
    np.testing.assert_allclose(np.array(m.dof_invweight0[:nv]), g[f'{key}__dof_invweight0'], rtol=1e-8)
    np.testing.assert_allclose(np.array([m.body_invweight0[b][0] for b in range(m.nbody)]), g[f'{key}__body_invweight0'][:, 0], rtol=1e-8, atol=1e-14)
    assert abs(m.meaninertia - float(g[f'{key}__meaninertia'])) < 1e-8 * m.meaninertia
    assert abs(float(g[f'{key}__timestep']) - m.timestep) < 1e-15
    # mj_forward on random states: same active set, accelerations to the solver tolerance
    q, v, w, u = (g[f'{key}__fwd_{k}'] for k in ('qpos', 'qvel', 'warm', 'ctrl'))
    for i in range(q.shape[1]):
        r = oracle.probe_forward(m, q[:, i], v[:, i], ctrl=u[:, i], warm=w[:, i])
        assert r['ncon'] == g[f'{key}__fwd_ncon'][i] and r['nefc'] == g[f'{key}__fwd_nefc'][i], i
        np.testing.assert_allclose(r['qacc'][:nv], g[f'{key}__fwd_qacc'][:, i], rtol=1e-6, atol=1e-6 * (1 + np.abs(g[f'{key}__fwd_qacc'][:, i]).max()))
    # mj_step from MuJoCo's own states with MuJoCo's warm-start schedule: one step ahead to integration accuracy
    pre, post = g[f'{key}__roll_pre'], g[f'{key}__roll_post']
    oracle.set_warmstart_schedule(1)
    try:
        for t in range(len(pre)):
            q1, v1, w1, rc = oracle.probe_steps(m, pre[t, 0].copy(), pre[t, 1].copy(), warm=pre[t, 2].copy(), n=1)
            assert rc == 0
            np.testing.assert_allclose(q1, post[t, 0], rtol=0, atol=1e-9)
            np.testing.assert_allclose(v1, post[t, 1], rtol=0, atol=1e-6)
            np.testing.assert_allclose(w1, post[t, 2], rtol=1e-5, atol=1e-5)
    finally:
        oracle.set_warmstart_schedule(0)


@pytest.mark.skipif(not os.path.exists(G12), reason=G12_SKIP)
@pytest.mark.parametrize('key', ['straight', 'walker165'])
def test_G12_oracle_matches_real_mujoco(oracle, key):
    check_g12(load('G12_mujoco_step.npz'), oracle, key)


def test_G12_harness_round_trip(oracle, tmp_path, monkeypatch):
    """The dump tool and the consumer above agree on the file format and on the state generator: the tool is run with the ORACLE
    standing in for the MuJoCo binding (so this pins nothing -- it only shows that the harness works the day a real dump arrives)."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location('dump_mujoco_vectors', os.path.join(ROOT, 'tools', 'dump_mujoco_vectors.py'))
    tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
    by_file = {v[0]: k for k, v in tool.MODELS.items()}

    class StandIn:
        def __init__(self, xml):
            self.model = g12_models()[by_file[os.path.basename(xml)]]
            m = self.model
            self.m = types.SimpleNamespace(nv=m.nv, nu=m.nu, nq=m.nv, qpos0=np.array(m.jnt_qpos0[:m.nv]), opt=types.SimpleNamespace(timestep=m.timestep, integrator=1))
            self.d = types.SimpleNamespace()
            self.version = 'oracle stand-in'

        def set(self, q, v, u, w):
            d = self.d
            d.qpos, d.qvel, d.ctrl, d.qacc_warmstart = np.array(q, float), np.array(v, float), np.array(u, float), np.array(w, float)

        def forward(self):
            d, nv = self.d, self.model.nv
            r = oracle.probe_forward(self.model, d.qpos, d.qvel, ctrl=d.ctrl, warm=d.qacc_warmstart)
            d.qacc, d.qfrc_constraint, d.ncon, d.nefc = r['qacc'][:nv], r['qfrc_constraint'][:nv], r['ncon'], r['nefc']

        def step(self):
            d = self.d
            oracle.set_warmstart_schedule(1)
            try:
                d.qpos, d.qvel, d.qacc_warmstart, rc = oracle.probe_steps(self.model, d.qpos, d.qvel, ctrl=d.ctrl, warm=d.qacc_warmstart, n=1)
            finally:
                oracle.set_warmstart_schedule(0)
            assert rc == 0

        def consts(self):
            m = self.model
            return np.array([list(m.body_invweight0[b]) for b in range(m.nbody)]), np.array(m.dof_invweight0[:m.nv]), m.meaninertia

    monkeypatch.setattr(tool, 'backend', StandIn)
    out = str(tmp_path / 'g12.npz')
    monkeypatch.setattr(sys, 'argv', ['dump_mujoco_vectors.py', '--out', out, '--states', '6', '--steps', '25'])
    tool.main()
    with np.load(out) as z:
        g = {k: z[k] for k in z.files}
    for key in ('straight', 'walker165'):
        check_g12(g, oracle, key)


def test_G12_absence_is_reported():
    """Keeps the pin status visible in every CPU run: the test passes either way, its output says which."""
    if os.path.exists(G12):
        print('G12 present: dynamics oracle pinned to', str(np.load(G12)['version']))
    else:
        import warnings
        warnings.warn(G12_SKIP)


def test_G7_monitor_lists():
    """Monitor's per-episode lists (rsi_positions, et_positions, difficult_rsi_phases, median_abs_torque_smoothed) as the host keeps
    them (drloco_amd/monitor_lists.py) from the four per-walker words of the step kernel, against the reference's own Monitor (G7)."""
    from drloco_amd.monitor_lists import MonitorLists
    g = load('G7_monitor.npz')
    T = len(g['rew'])
    ml = MonitorLists(1, ep_dur_max=3000)
    # the device words, restated: cursor position at the first step / at the end of an episode, the step's torque, the flag
    ep_len, init_pos = 0, 0
    for t in range(T):
        if ep_len == 0:
            init_pos = int(g['cursor'][t])
        ep_len += 1
        done = bool(g['done'][t])
        difficult = float(done and ep_len < g['ep_len_smoothed'][t] * 0.75)
        ml.update(np.array([done]), np.array([init_pos]), np.array([g['cursor'][t]]), np.array([g['tor'][t]]), np.array([difficult]))
        np.testing.assert_allclose(ml.median_abs_torque_smoothed[0], g['median_abs_torque_smoothed'][t], rtol=1e-13)
        if done:
            ep_len = 0
    assert ml.rsi_positions[0] == list(g['rsi_positions']) and ml.et_positions[0] == list(g['et_positions'])
    assert ml.difficult_rsi_phases[0] == list(g['difficult_rsi_phases']) and len(g['difficult_rsi_phases']) > 0


def g15_walk(env, g, refs, per_step, per_reset, to_f64=lambda x: x):
    """Drive `env` (oracle or device wrapper with the oracle's method names) through golden G15's episodes: the injected RSI draws / evaluation inits, the injected
    end states of every control step, the falls that end the episodes; calls per_reset(e, obs0, state) after every reset and per_step(t, e, obs, rew, done, term, terms)."""
    ep_len, mode, draw = g['ep_len'], g['ep_mode'], g['ep_draw']

    def arm(e):          # what the next reset will do
        env.set_eval(bool(mode[e]))
        if not mode[e]:
            env.inject_rsi(0, int(draw[e, 0]), int(draw[e, 1]))
    arm(0)
    obs0 = env.reset()
    t = 0
    for e in range(len(ep_len)):
        per_reset(e, obs0[0])
        for k in range(int(ep_len[e])):
            if k == ep_len[e] - 1 and e + 1 < len(ep_len):
                arm(e + 1)
            env.inject_state(0, g['inj_q'][t], g['inj_v'][t])
            obs, rew, done, term, terms = env.step(g['actions'][t][None])
            per_step(t, e, obs[0], rew[0], bool(done[0]), term[0], terms[0])
            t += 1
        obs0 = obs          # the auto-reset's observation opens the next episode


def test_G15_quirk_Q4_com_z_offsets_persist_in_the_data_set(oracle, model, refs):
    """Quirk Q4 (base_ref_trajecs.py:126-127 via mimic_env.py:555-557): golden G15 is ONE reference environment living through nine episodes -- two more resets onto
    a step an earlier episode started on, one onto a step an earlier episode rolled through, two evaluation inits (which mutate step 0's row, quirk Q3), a reset
    onto step 0 afterwards, a wrap 29 -> 0.  The oracle follows every reference COM-z value, reward term, the reward under a weight vector with a COM weight,
    the initial states and the Monitor's smoothed COM term."""
    g = load('G15_q4_com_z.npz')
    env = make_env(oracle, model, refs, n=1, rew_weights=list(g['weights']))
    zrow_pristine = lambda s: refs.table[2, refs.step_off[s]:refs.step_off[s + 1]]

    def per_reset(e, obs0):
        st = env.get_state()
        np.testing.assert_allclose(st['qpos'][:, 0], g['ep_qpos0'][e], rtol=0, atol=1e-14)
        assert np.array_equal(st['qvel'][:, 0], g['ep_qvel0'][e])
        np.testing.assert_allclose(obs0, g['ep_obs0'][e], rtol=0, atol=1e-14)
        cur = st['cursor'][:, 0]
        assert (cur[abi.DL_CUR_I_STEP], cur[abi.DL_CUR_POS]) == tuple(g['ep_cursor0'][e][:2]) and cur[abi.DL_CUR_READ_STEP] == g['ep_read_step'][e]
        # the row adjust_COM_Z_pos mutated, as the reference left it: pristine - accumulated offset
        s = int(g['ep_read_step'][e]); L = refs.step_len[s]
        z = zrow_pristine(s)[[0, 1, L // 2, L - 1]] - env.get_ref_offsets()[s, 0]
        np.testing.assert_allclose(z, g['ep_zrow'][e], rtol=0, atol=1e-14)

    seen = dict(com_min=1.0)

    def per_step(t, e, obs, rew, done, term, terms):
        assert done == bool(g['done'][t]), t
        if done:
            assert rew == 0 and np.signbit(rew)          # a fall: -0.0
            np.testing.assert_allclose(term, g['obs'][t], rtol=0, atol=1e-14)
        else:
            np.testing.assert_allclose(obs, g['obs'][t], rtol=0, atol=1e-14)
            qr, vr = env.ref_lookup(0)
            np.testing.assert_allclose(qr, g['ref_q'][t], rtol=0, atol=1e-14, err_msg=f'step {t} episode {e}')          # COM-z with the offsets of Q4, COM-x with the offset of Q1
            assert np.array_equal(vr, g['ref_v'][t])
            np.testing.assert_allclose(terms, g['terms'][t], rtol=1e-12)
            assert abs(rew - g['rew'][t]) <= 1e-12
            seen['com_min'] = min(seen['com_min'], terms[2])
        assert abs(env.stats('mean_ep_com_rew_smoothed')[0] - g['mean_ep_com_rew_smoothed'][t]) <= 1e-12, t
        assert abs(env.stats('mean_ep_pos_rew_smoothed')[0] - g['mean_ep_pos_rew_smoothed'][t]) <= 1e-12, t

    g15_walk(env, g, refs, per_step, per_reset)
    assert seen['com_min'] < 0.9          # the COM term is not trivially 1 in the fixture
    # ... and the fixture really tells the two behaviours apart: with the per-episode offset of rounds 1-5 the COM reference differs after the first rollover
    env2 = make_env(oracle, model, refs, n=1, rew_weights=list(g['weights']), intended_semantics=abi.DL_INTENDED_COMZ_PER_EPISODE)
    worst = [0.0]

    def per_step2(t, e, obs, rew, done, term, terms):
        if not done:
            worst[0] = max(worst[0], abs(env2.ref_lookup(0)[0][2] - g['ref_q'][t][2]))
    g15_walk(env2, g, refs, per_step2, lambda e, o: None)
    assert worst[0] > 5e-3


def test_intended_semantics_switches(oracle, model, refs):
    """dl_config.intended_semantics: Q2 off restarts count_steps_same_vel with every reset, Q3 off lets an evaluation init read its own step's table."""
    env = make_env(oracle, model, refs, n=1, intended_semantics=abi.DL_INTENDED_COUNT_PER_EPISODE | abi.DL_INTENDED_EVAL_OWN_STEP)
    c = cursor(3, 10, count=17)
    env.set_state(cursor=c)
    env.inject_rsi(0, 8, 20)
    env.reset()
    cur = env.get_state()['cursor'][:, 0]
    assert cur[abi.DL_CUR_COUNT] == 1 and cur[abi.DL_CUR_I_STEP] == 8
    env.inject_rsi(0, -1, 0)
    env.set_eval(True)
    env.set_state(cursor=cursor(0, 0) + np.eye(abi.DL_CUR_WORDS, 1, -abi.DL_CUR_EVAL_K, dtype=np.int32) * 4)          # next evaluation init: k = 4
    env.reset()
    st = env.get_state()
    cur = st['cursor'][:, 0]
    assert cur[abi.DL_CUR_I_STEP] == 4 and cur[abi.DL_CUR_READ_STEP] == 4
    p = int(0.75 * refs.step_len[4])
    want = refs.table[:14, refs.step_off[4] + p]
    assert np.array_equal(np.delete(st['qpos'][:, 0], 2), np.delete(want, 2))
