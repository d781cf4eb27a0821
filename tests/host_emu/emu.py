"""ctypes wrapper of the host-emulation build of the kernel source (TEST INFRASTRUCTURE)."""
import ctypes as C
import os
import subprocess

import numpy as np

from drloco_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_SAN = os.environ.get('DL_EMU_SANITIZE') == '1'          # UBSan build of the kernel source on the host (tests/test_sanitizers.py); ASan and the fibers' hand-made stacks do not mix
_R4BUG = os.environ.get('DL_EMU_R4BUG') == '1'          # -DDL_EXP_R4_LATE_READ: the round-4 hand-over defect re-introduced (tests/test_split_protocol_emu.py shows that the schedules catch it)
_LIB = os.path.join(_HERE, 'libdl_emu_r4bug.so' if _R4BUG else ('libdl_emu_ubsan.so' if _SAN else 'libdl_emu.so'))
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, 'emu.cpp'), os.path.join(_HERE, 'dl_group_emu.hpp')] \
        + [os.path.join(_ROOT, 'drloco_amd', 'csrc', f) for f in ('dl_core.hpp', 'dl_env.hpp', 'dl_host.hpp', 'dl_group.hpp', 'dl_group_env.hpp')] \
        + [os.path.join(_ROOT, 'include', 'drloco_hip.h')]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(['g++', '-O1' if _SAN else '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off'] + (['-fsanitize=undefined', '-fno-sanitize-recover=all'] if _SAN else []) + (['-DDL_EXP_R4_LATE_READ', '-DDL_EMU_ONLY_F32'] if _R4BUG else []) + [
                               '-I' + os.path.join(_ROOT, 'include'), '-I' + os.path.join(_ROOT, 'drloco_amd', 'csrc'), '-I' + _HERE,
                               '-o', _LIB, srcs[0]])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for suf in (('f32',) if _R4BUG else ('f32', 'f64', 'f32_165', 'f64_165')):
            getattr(_lib, 'dle_create_' + suf).restype = C.c_void_p
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class EmuEnv:
    def __init__(self, model, refs, cfg, n_envs, precision=64):
        self.suf = ('f64' if precision == 64 else 'f32') + ('_165' if cfg.env_kind == abi.DL_ENV_LOCO3D else '')
        self.rt = np.float64 if precision == 64 else np.float32
        self.ct = C.c_double if precision == 64 else C.c_float
        self.n, self.nv, self.nu = n_envs, model.nv, model.nu
        self.obs_dim = (10 + 2 * model.nv - 1) if cfg.env_kind == abi.DL_ENV_LOCO3D else 2 * model.nv + 1
        self._desc = refs.as_desc()
        self.h = C.c_void_p(self._f('create')(C.byref(model), C.byref(self._desc), C.byref(cfg), n_envs))
        assert self.h

    def _f(self, name):
        return getattr(lib(), f'dle_{name}_{self.suf}')

    def close(self):
        if self.h:
            self._f('destroy')(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self, mask=None, init_step=None, init_pos=None):
        obs = np.zeros((self.n, self.obs_dim), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        s = None if init_step is None else np.ascontiguousarray(init_step, np.int32)
        p = None if init_pos is None else np.ascontiguousarray(init_pos, np.int32)
        self._f('reset')(self.h, _p(m, C.c_uint8), _p(s, C.c_int32), _p(p, C.c_int32), _p(obs, C.c_float))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32)
        obs = np.zeros((self.n, self.obs_dim), np.float32); term = np.zeros_like(obs)
        rew = np.zeros(self.n, np.float32); done = np.zeros(self.n, np.uint8); terms = np.zeros((self.n, 3), np.float32)
        self._f('step')(self.h, _p(a, C.c_float), _p(obs, C.c_float), _p(rew, C.c_float), _p(done, C.c_uint8), _p(term, C.c_float), _p(terms, C.c_float))
        return obs, rew, done, term, terms

    def get_state(self):
        q = np.zeros((self.nv, self.n), self.rt); v = np.zeros_like(q); w = np.zeros_like(q)
        cur = np.zeros((abi.DL_CUR_WORDS, self.n), np.int32); walked = np.zeros(self.n)
        self._f('get_state')(self.h, _p(q, self.ct), _p(v, self.ct), _p(w, self.ct), _p(cur, C.c_int32), _p(walked, C.c_double))
        return dict(qpos=q, qvel=v, warm=w, cursor=cur, walked=walked)

    def set_state(self, qpos=None, qvel=None, warm=None, cursor=None, walked=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, self.rt)
        q, v, w = f(qpos), f(qvel), f(warm)
        c = None if cursor is None else np.ascontiguousarray(cursor, np.int32)
        wk = None if walked is None else np.ascontiguousarray(walked, np.float64)
        self._f('set_state')(self.h, _p(q, self.ct), _p(v, self.ct), _p(w, self.ct), _p(c, C.c_int32), _p(wk, C.c_double))

    def forward(self, ctrl=None):
        u = np.zeros((self.nu, self.n), self.rt) if ctrl is None else np.ascontiguousarray(ctrl, self.rt)
        qacc = np.zeros((self.nv, self.n), self.rt)
        ncon = np.zeros(self.n, np.int32); nefc = np.zeros(self.n, np.int32); nit = np.zeros(self.n, np.int32)
        self._f('forward')(self.h, _p(u, self.ct), _p(qacc, self.ct), _p(ncon, C.c_int32), _p(nefc, C.c_int32), _p(nit, C.c_int32))
        return qacc, ncon, nefc, nit

    # ---- the 16-lanes-per-walker kernels (dl_group.hpp / dl_group_env.hpp), a wave as 64 fibers
    def gforward(self, ctrl=None):
        u = np.zeros((self.nu, self.n), self.rt) if ctrl is None else np.ascontiguousarray(ctrl, self.rt)
        qacc = np.zeros((self.nv, self.n), self.rt)
        ncon = np.zeros(self.n, np.int32); nefc = np.zeros(self.n, np.int32); nit = np.zeros(self.n, np.int32)
        rc = self._f('gforward')(self.h, _p(u, self.ct), _p(qacc, self.ct), _p(ncon, C.c_int32), _p(nefc, C.c_int32), _p(nit, C.c_int32))
        assert rc == 0
        return qacc, ncon, nefc, nit

    def gstep(self, actions, return_ctrl=False):
        """actions [N, nu] (one control step, term_obs / reward terms returned) or [T, N, nu] (T steps in one 'launch')."""
        a = np.ascontiguousarray(actions, np.float32)
        multi = a.ndim == 3
        T = a.shape[0] if multi else 1
        obs = np.zeros((T, self.n, self.obs_dim), np.float32); term = np.zeros((T, self.n, self.obs_dim), np.float32)
        rew = np.zeros((T, self.n), np.float32); done = np.zeros((T, self.n), np.uint8); terms = np.zeros((T, self.n, 3), np.float32)
        ctrl = np.zeros((T, self.n, self.nu), np.float32)
        rc = self._f('gstep')(self.h, C.c_int(T), _p(a, C.c_float), _p(obs, C.c_float), _p(rew, C.c_float), _p(done, C.c_uint8), _p(term, C.c_float), _p(terms, C.c_float), _p(ctrl, C.c_float))
        assert rc == 0
        out = (obs, rew, done, term, terms) if multi else (obs[0], rew[0], done[0], term[0], terms[0])
        return out + ((ctrl if multi else ctrl[0]),) if return_ctrl else out

    def gstep_split(self, actions, policy=0, seed=0):
        """The split workgroup's wave pair (dynamics wave + look-ahead partner, drloco_amd/csrc/dl_group_env.hpp) as two emulated waves sharing the pair's LDS;
        `policy` / `seed` select how their rounds are interleaved (dl_group_emu.hpp run_pair).  actions [T, N, nu] or [N, nu]; returns (obs, rew, done, fault bits)."""
        a = np.ascontiguousarray(actions, np.float32)
        multi = a.ndim == 3
        T = a.shape[0] if multi else 1
        obs = np.zeros((T, self.n, self.obs_dim), np.float32); rew = np.zeros((T, self.n), np.float32); done = np.zeros((T, self.n), np.uint8)
        f = self._f('gstep_split')
        f.restype = C.c_int
        rc = f(self.h, C.c_int(T), _p(a, C.c_float), _p(obs, C.c_float), _p(rew, C.c_float), _p(done, C.c_uint8), C.c_int(policy), C.c_uint64(seed))
        assert rc >= 0
        return (obs, rew, done, rc) if multi else (obs[0], rew[0], done[0], rc)

    def set_randomization(self, mass_scale=None, floor_friction=None, push=None):
        f = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)
        ms, mu, pu = f(mass_scale), f(floor_friction), f(push)
        self._f('set_rnd')(self.h, _p(ms, C.c_float), _p(mu, C.c_float), _p(pu, C.c_float))

    def set_push_schedule(self, force, phase, period, duration):
        f = np.ascontiguousarray(force, np.float32); ph = np.ascontiguousarray(phase, np.int32)
        self._f('set_push_schedule')(self.h, _p(f, C.c_float), _p(ph, C.c_int32), C.c_int(period), C.c_int(duration))

    def glds_bytes(self):
        return self._f('glds_bytes')()

    def inject_state(self, i, qpos, qvel):
        q = np.ascontiguousarray(qpos, self.rt); v = np.ascontiguousarray(qvel, self.rt)
        self._f('inject')(self.h, C.c_int(i), C.c_int(1), _p(q, self.ct), _p(v, self.ct))

    def get_ref_offsets(self):
        """quirk Q4's record (DevState::zacc): [n_steps, N]"""
        z = np.zeros((self._desc.n_steps, self.n), self.rt)
        self._f('get_zacc')(self.h, _p(z, self.ct))
        return z

    def set_eval(self, on=True):
        self._f('set_eval')(self.h, C.c_int(int(on)))

    def inject_exception(self, i):
        self._f('inject')(self.h, C.c_int(i), C.c_int(2), None, None)

    def inject_rsi(self, i, step, pos):
        self._f('inject_rsi')(self.h, C.c_int(i), C.c_int(step), C.c_int(pos))
