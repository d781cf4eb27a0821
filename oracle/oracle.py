"""ctypes wrapper of the CPU oracle (oracle/libdl_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
drloco_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

from drloco_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get('DL_ORACLE_LIB') or os.path.join(_HERE, 'libdl_oracle.so')          # DL_ORACLE_LIB: the sanitizer build (make SAN=1), tests/test_sanitizers.py
DLO_MAXCON = 32
DLO_MAXEFC = 4 * DLO_MAXCON + 2 * abi.DL_MAX_DOF

F_NOCONTACT, F_NOLIMIT, F_NODAMP, F_NOGRAV, F_NOACT = 1, 2, 4, 8, 16


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ('dl_oracle.c', 'dl_oracle.h')] + [os.path.join(_HERE, '..', 'include', 'drloco_hip.h')]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(['make', '-s', '-C', _HERE, '-B'] + (['SAN=1'] if _LIB.endswith('_san.so') else []))
    return _LIB


class Probe(C.Structure):
    _d, _i = C.c_double, C.c_int32
    _fields_ = [
        ('M', _d * (abi.DL_MAX_DOF * abi.DL_MAX_DOF)), ('qfrc_bias', _d * abi.DL_MAX_DOF),
        ('qfrc_smooth', _d * abi.DL_MAX_DOF), ('qacc_smooth', _d * abi.DL_MAX_DOF), ('qacc', _d * abi.DL_MAX_DOF),
        ('qfrc_constraint', _d * abi.DL_MAX_DOF), ('xpos', _d * (abi.DL_MAX_BODY * 3)),
        ('xmat', _d * (abi.DL_MAX_BODY * 9)), ('xipos', _d * (abi.DL_MAX_BODY * 3)),
        ('site_xpos', _d * (abi.DL_MAX_SITE * 3)), ('energy', _d * 2),
        ('ncon', _i), ('nefc', _i), ('niter', _i),
        ('con_pos', _d * (DLO_MAXCON * 3)), ('con_dist', _d * DLO_MAXCON), ('con_frame', _d * (DLO_MAXCON * 9)),
        ('con_geom', _i * DLO_MAXCON), ('efc_J', _d * (DLO_MAXEFC * abi.DL_MAX_DOF)),
        ('efc_pos', _d * DLO_MAXEFC), ('efc_D', _d * DLO_MAXEFC), ('efc_aref', _d * DLO_MAXEFC),
        ('efc_force', _d * DLO_MAXEFC), ('solver_cost', _d),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.dlo_create.restype = C.c_void_p
        _lib.dlo_create.argtypes = [C.POINTER(abi.ModelDesc), C.POINTER(abi.RefsDesc), C.POINTER(abi.Config), C.c_int32]
        _lib.dlo_probe_steps.restype = C.c_int
        _lib.dlo_stats_snapshot.restype = C.c_int
    return _lib


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def set_warmstart_schedule(schedule):
    """0: every forward evaluation overwrites qacc_warmstart (device kernels); 1: once per mj_step (MuJoCo 2.x)."""
    lib().dlo_set_warmstart_schedule(C.c_int(int(schedule)))


def set_const(model):
    lib().dlo_set_const(C.byref(model))
    return model


def probe_forward(model, qpos, qvel, ctrl=None, warm=None, flags=0):
    out = Probe()
    q = np.ascontiguousarray(qpos, np.float64)
    v = np.ascontiguousarray(qvel, np.float64)
    u = None if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
    w = None if warm is None else np.ascontiguousarray(warm, np.float64)
    lib().dlo_probe_forward(C.byref(model), _p(q), _p(v), _p(u), _p(w), C.c_int(flags), C.byref(out))
    nv, nb = model.nv, model.nbody
    r = dict(M=np.array(out.M[:nv * nv]).reshape(nv, nv), ncon=out.ncon, nefc=out.nefc, niter=out.niter,
             energy=np.array(out.energy[:]), solver_cost=out.solver_cost)
    for k in ('qfrc_bias', 'qfrc_smooth', 'qacc_smooth', 'qacc', 'qfrc_constraint'):
        r[k] = np.array(getattr(out, k)[:nv])
    r['xpos'] = np.array(out.xpos[:3 * nb]).reshape(nb, 3)
    r['xmat'] = np.array(out.xmat[:9 * nb]).reshape(nb, 3, 3)
    r['xipos'] = np.array(out.xipos[:3 * nb]).reshape(nb, 3)
    r['site_xpos'] = np.array(out.site_xpos[:3 * model.nsite]).reshape(model.nsite, 3)
    r['con_pos'] = np.array(out.con_pos[:3 * out.ncon]).reshape(out.ncon, 3)
    r['con_dist'] = np.array(out.con_dist[:out.ncon])
    r['con_frame'] = np.array(out.con_frame[:9 * out.ncon]).reshape(out.ncon, 3, 3)
    r['con_geom'] = np.array(out.con_geom[:out.ncon])
    r['efc_J'] = np.array(out.efc_J[:out.nefc * nv]).reshape(out.nefc, nv)
    for k in ('efc_pos', 'efc_D', 'efc_aref', 'efc_force'):
        r[k] = np.array(getattr(out, k)[:out.nefc])
    return r


def probe_steps(model, qpos, qvel, ctrl=None, warm=None, dt=None, n=1, flags=0):
    """n RK4 mj_steps; returns (qpos, qvel, warm, diverged_at)."""
    q = np.array(qpos, np.float64)
    v = np.array(qvel, np.float64)
    u = np.zeros(model.nu) if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
    w = np.zeros(model.nv) if warm is None else np.array(warm, np.float64)
    rc = lib().dlo_probe_steps(C.byref(model), _p(q), _p(v), _p(u), _p(w), C.c_double(dt or model.timestep), C.c_int(n), C.c_int(flags))
    return q, v, w, rc


class OracleEnv:
    """N walkers on the CPU oracle; mirrors the dl_* C-ABI with host numpy arrays."""

    def __init__(self, model, refs, cfg, n_envs):
        self.model, self.refs, self.cfg, self.n = model, refs, cfg, n_envs
        self._desc = refs.as_desc()
        self.h = C.c_void_p(lib().dlo_create(C.byref(model), C.byref(self._desc), C.byref(cfg), n_envs))
        self.nv, self.nu = model.nv, model.nu
        self.obs_dim = (10 + 2 * self.nv - 1) if cfg.env_kind == abi.DL_ENV_LOCO3D else (1 + 1 + (self.nv - 1) + self.nv)

    def close(self):
        if self.h:
            lib().dlo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self, mask=None, init_step=None, init_pos=None):
        obs = np.zeros((self.n, self.obs_dim))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        s = None if init_step is None else np.ascontiguousarray(init_step, np.int32)
        p = None if init_pos is None else np.ascontiguousarray(init_pos, np.int32)
        lib().dlo_reset(self.h, _p(m, C.c_uint8), _p(s, C.c_int32), _p(p, C.c_int32), _p(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float64)
        assert a.shape == (self.n, self.nu)
        obs = np.zeros((self.n, self.obs_dim))
        term = np.zeros((self.n, self.obs_dim))
        rew = np.zeros(self.n)
        done = np.zeros(self.n, np.uint8)
        terms = np.zeros((self.n, 3))
        lib().dlo_step(self.h, _p(a), _p(obs), _p(rew), _p(done, C.c_uint8), _p(term), _p(terms))
        return obs, rew, done, term, terms

    def get_state(self):
        q = np.zeros((self.nv, self.n)); v = np.zeros((self.nv, self.n)); w = np.zeros((self.nv, self.n))
        cur = np.zeros((abi.DL_CUR_WORDS, self.n), np.int32)
        walked = np.zeros(self.n)
        lib().dlo_get_state(self.h, _p(q), _p(v), _p(w), _p(cur, C.c_int32), _p(walked))
        return dict(qpos=q, qvel=v, warm=w, cursor=cur, walked=walked)

    def set_state(self, qpos=None, qvel=None, warm=None, cursor=None, walked=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        q, v, w, wk = f(qpos), f(qvel), f(warm), f(walked)
        c = None if cursor is None else np.ascontiguousarray(cursor, np.int32)
        lib().dlo_set_state(self.h, _p(q), _p(v), _p(w), _p(c, C.c_int32), _p(wk))

    def get_ref_offsets(self):
        z = np.zeros((self.refs.n_steps, self.n))
        lib().dlo_get_ref_offsets(self.h, _p(z))
        return z

    def set_ref_offsets(self, z):
        z = np.ascontiguousarray(z, np.float64)
        assert z.shape == (self.refs.n_steps, self.n)
        lib().dlo_set_ref_offsets(self.h, _p(z))

    def forward(self, ctrl=None):
        u = np.zeros((self.nu, self.n)) if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
        qacc = np.zeros((self.nv, self.n))
        ncon = np.zeros(self.n, np.int32); nefc = np.zeros(self.n, np.int32); nit = np.zeros(self.n, np.int32)
        lib().dlo_forward(self.h, _p(u), _p(qacc), _p(ncon, C.c_int32), _p(nefc, C.c_int32), _p(nit, C.c_int32))
        return qacc, ncon, nefc, nit

    def set_eval(self, on=True):
        lib().dlo_set_eval(self.h, C.c_int32(int(on)))

    def inject_exception(self, i):
        lib().dlo_inject_exception(self.h, C.c_int32(i))

    def inject_rsi(self, i, step, pos):
        lib().dlo_inject_rsi(self.h, C.c_int32(i), C.c_int32(step), C.c_int32(pos))

    def inject_state(self, i, qpos, qvel):
        q = np.ascontiguousarray(qpos, np.float64); v = np.ascontiguousarray(qvel, np.float64)
        lib().dlo_inject_state(self.h, C.c_int32(i), _p(q), _p(v))

    def observe(self):
        obs = np.zeros((self.n, self.obs_dim)); imit = np.zeros(self.n); terms = np.zeros((self.n, 3))
        lib().dlo_observe(self.h, _p(obs), _p(imit), _p(terms))
        return obs, imit, terms

    def last_ctrl(self):
        out = np.zeros((self.n, self.nu))
        lib().dlo_last_ctrl(self.h, _p(out))
        return out

    def monitor_feed(self, i, rew, done, pos, vel, com, tor, walked):
        lib().dlo_monitor_feed(self.h, C.c_int32(i), C.c_double(rew), C.c_int32(int(done)), C.c_double(pos),
                               C.c_double(vel), C.c_double(com), C.c_double(tor), C.c_double(walked))

    def stats(self, name):
        out = np.zeros(self.n)
        if lib().dlo_stats_snapshot(self.h, name.encode(), _p(out)) != 0:
            raise KeyError(name)
        return out

    def ref_lookup(self, i):
        q = np.zeros(self.nv); v = np.zeros(self.nv)
        lib().dlo_ref_lookup(self.h, C.c_int32(i), _p(q), _p(v))
        return q, v

    def set_randomization(self, mass_scale=None, floor_friction=None, xfrc=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        ms, fr, xf = f(mass_scale), f(floor_friction), f(xfrc)
        lib().dlo_set_randomization(self.h, _p(ms), _p(fr), _p(xf))

    def terminate_early(self, i):
        f = np.zeros(4, np.int32)
        lib().dlo_terminate_early(self.h, C.c_int32(i), _p(f, C.c_int32))
        return f


def moments_update(mean, var, count, x):
    x = np.ascontiguousarray(x, np.float64)
    cnt = np.array([count], np.float64)
    lib().dlo_moments_update(_p(mean), _p(var), _p(cnt), _p(x), C.c_int32(x.shape[0]), C.c_int32(x.shape[1]))
    return float(cnt[0])


def gae(rew, val, ep_start, last_val, last_done, gamma, lam):
    T, N = rew.shape
    f = lambda a: np.ascontiguousarray(a, np.float32)
    rew, val, last_val = f(rew), f(val), f(last_val)
    es = np.ascontiguousarray(ep_start, np.uint8); ld = np.ascontiguousarray(last_done, np.uint8)
    adv = np.zeros((T, N), np.float32); ret = np.zeros((T, N), np.float32)
    lib().dlo_gae(_p(rew, C.c_float), _p(val, C.c_float), _p(es, C.c_uint8), _p(last_val, C.c_float), _p(ld, C.c_uint8),
                  C.c_float(gamma), C.c_float(lam), C.c_int32(T), C.c_int32(N), _p(adv, C.c_float), _p(ret, C.c_float))
    return adv, ret


def rsi_draw(seed, global_env, episode, step_off):
    so = np.ascontiguousarray(step_off, np.int32)
    s = C.c_int32(); p = C.c_int32()
    lib().dlo_rsi_draw(C.c_uint64(seed), C.c_uint32(global_env), C.c_uint32(episode), C.c_int32(len(so) - 1), _p(so, C.c_int32), C.byref(s), C.byref(p))
    return s.value, p.value


def policy_forward(w1, b1, w2, b2, wa, ba, wv, bv, log_std, obs, eps):
    """float64 numpy restatement of SB3 1.0 ActorCriticPolicy.forward for the reference's CustomActorCriticPolicy
    (/root/reference/drloco/custom/policies.py:13-51: one tanh trunk shared by policy_net and value_net;
    action_net / value_net heads; DiagGaussianDistribution sample + log_prob).  Returns (latent, actions, values, log_probs)."""
    f = lambda a: np.asarray(a, np.float64)
    h = np.tanh(f(obs) @ f(w1).T + f(b1))
    lat = np.tanh(h @ f(w2).T + f(b2))
    mean = lat @ f(wa).T + f(ba)
    val = (lat @ f(wv).T + f(bv))[:, 0]
    e = f(eps)
    act = mean + np.exp(f(log_std)) * e
    logp = (-0.5 * e ** 2 - f(log_std) - 0.5 * np.log(2 * np.pi)).sum(1)
    return lat, act, val, logp
