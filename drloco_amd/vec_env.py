"""HipVecEnv / HipVecNormalize: the reference's `utils.vec_env()` product
(VecNormalize(SubprocVecEnv([Monitor(MimicEnv)]*n)), drloco/common/utils.py:97-134) as one
object over N walkers resident on one MI355X.

The classes are duck-typed after stable_baselines3 1.0's VecEnv / VecNormalize (SB3 and gym are
not installed in this image): num_envs, observation_space, action_space, reset, step_async,
step_wait, step, get_attr, set_attr, env_method, seed, close; VecNormalize adds obs_rms,
ret_rms, normalize_obs, get_original_obs/reward, save/load and `.venv`.

`step()` follows the VecEnv contract and returns numpy arrays (one D2H copy per control step).
`step_tensors()` keeps everything on the device for GPU-resident rollouts."""
import ctypes as C
import pickle

import numpy as np
import torch

from . import abi, compat, lib, mocap, models, monitor_lists

MONITOR_ATTRS = ('ep_len_smoothed', 'ep_ret_smoothed', 'mean_reward_smoothed', 'moved_distance',
                 'mean_ep_pos_rew_smoothed', 'mean_ep_vel_rew_smoothed', 'mean_ep_com_rew_smoothed',
                 'mean_abs_ep_torque_smoothed')


Box = compat.MiniBox          # kept under its old name for callers that built spaces by hand
_VecEnvBase, _VecEnvWrapperBase = compat.VecEnvBase, compat.VecEnvWrapperBase


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class HipVecEnv(_VecEnvBase):
    """N MimicWalker3dEnv walkers stepped by the HIP kernels (one process per GPU).  An SB3 `VecEnv` when stable-baselines3
    imports (drloco_amd/compat.py), so that `PPO(policy, env)` takes it as it is (train.py:110)."""

    def __init__(self, env_id=models.STRAIGHT_WALKER, num_envs=4096, device=None, seed=33, precision=32,
                 model=None, refs=None, env_index_base=0, **config):
        if not torch.cuda.is_available():
            raise lib.DrlocoError('HipVecEnv needs a HIP device; there is no CPU fallback')
        split = config.get('lanes_per_walker') == 'split'          # shorthand: 16 lanes per walker + set_split(True)
        if split:
            config['lanes_per_walker'] = 16
        self._lib = lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.env_id = env_id
        self.model = model if model is not None else models.make_model(env_id)
        kind = models.ENV_KIND[env_id]
        if refs is None:
            if kind == abi.DL_ENV_LOCO3D:
                raise lib.DrlocoError('MimicWalker165cm65kg needs a reference table: pass refs=mocap.convert_loco3d_mat(path) '
                                      '(loco3d_guoping.mat is not part of the reference checkout) or a synthetic one')
            refs = mocap.RefTable.load()
        self.refs = refs
        make_cfg = abi.loco3d_config if kind == abi.DL_ENV_LOCO3D else abi.default_config
        self.cfg = make_cfg(seed=seed, precision=precision, env_index_base=env_index_base, **config)
        self.num_envs = int(num_envs)
        self.precision = precision
        self.rdtype = torch.float64 if precision == 64 else torch.float32
        h = C.c_void_p()
        desc = self.refs.as_desc()
        with torch.cuda.device(self.device):
            lib.check(self._lib.dl_create(C.byref(self.model), C.byref(desc), C.byref(self.cfg), self.num_envs,
                                          self.device.index, C.byref(h)))
        self._h = h
        self.nv, self.nu = self.model.nv, self.model.nu
        self.obs_dim = self._lib.dl_obs_dim(h)
        # MujocoEnv's spaces (gym 0.18 mujoco_env.py: observation Box(-inf, inf), action Box = the actuators' ctrlrange)
        obs_space = compat.make_box(np.full(self.obs_dim, -np.inf), np.full(self.obs_dim, np.inf))
        act_space = compat.make_box([self.model.act_ctrlrange[a][0] for a in range(self.nu)], [self.model.act_ctrlrange[a][1] for a in range(self.nu)])
        _VecEnvBase.__init__(self, self.num_envs, obs_space, act_space)
        n, dev = self.num_envs, self.device
        self.obs = torch.zeros(n, self.obs_dim, device=dev)
        self.rew = torch.zeros(n, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.term_obs = torch.zeros(n, self.obs_dim, device=dev)
        self.rew_terms = torch.zeros(n, 3, device=dev)
        self._actions = None
        # host-side mirrors of Monitor's per-env lists (monitor_wrapper.py:57,120)
        self._ep_len = np.zeros(n, np.int64)
        self.ep_lens = [[] for _ in range(n)]
        self._mlists = None                        # track_monitor_lists()
        self.reuse_infos, self._info_pool, self._info_dirty = True, None, []          # _infos()
        self.split = False
        if split:
            self.set_split(True)

    # ---- VecEnv surface -------------------------------------------------------------------
    def reset(self, mask=None, init_step=None, init_pos=None):
        self.reset_tensors(mask, init_step, init_pos)
        return self.obs.cpu().numpy()

    def reset_tensors(self, mask=None, init_step=None, init_pos=None):
        dev = self.device
        m = None if mask is None else torch.as_tensor(mask, dtype=torch.uint8, device=dev).contiguous()
        s = None if init_step is None else torch.as_tensor(init_step, dtype=torch.int32, device=dev).contiguous()
        p = None if init_pos is None else torch.as_tensor(init_pos, dtype=torch.int32, device=dev).contiguous()
        lib.check(self._lib.dl_reset(self._h, _ptr(m), _ptr(s), _ptr(p), _ptr(self.obs), _stream()))
        return self.obs

    def step_tensors(self, actions, done_out=None, obs_out=None, rew_out=None):
        """actions: float32 cuda tensor [N, nu]; returns device tensors (obs, rew, done, term_obs).
        done_out: optional uint8 [N] destination for the done flags (e.g. the next episode_starts slot);
        obs_out / rew_out: optional destinations of the raw observation / reward (default: self.obs / self.rew)."""
        a = actions.to(device=self.device, dtype=torch.float32).contiguous()
        assert a.shape == (self.num_envs, self.nu)
        done = self.done if done_out is None else done_out
        obs = self.obs if obs_out is None else obs_out
        rew = self.rew if rew_out is None else rew_out
        lib.check(self._lib.dl_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done),
                                    _ptr(self.term_obs), _ptr(self.rew_terms), _stream()))
        return obs, rew, done, self.term_obs

    def step_async(self, actions):
        self._actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32, device=self.device)

    def step_wait(self):
        obs, rew, done, term = self.step_tensors(self._actions)
        obs_h, rew_h, done_h = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)
        infos = self._infos()
        self._ep_len += 1
        self._update_monitor_lists(done_h)
        if done_h.any():
            term_h = term.cpu().numpy()
            for i in np.nonzero(done_h)[0]:
                infos[i]['terminal_observation'] = term_h[i]
                self._info_dirty.append(int(i))
                self.ep_lens[i].append(int(self._ep_len[i]))
                self._ep_len[i] = 0
        return obs_h, rew_h, done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _infos(self):
        """The step's list of info dicts.  MimicEnv.step returns `{}` (mimic_env.py:126) and the vec-env worker adds 'terminal_observation' for a finished env, so all
        but a few of a step's dicts are empty: every walker keeps ONE dict object across steps (building N fresh dicts per step costs more host time than the step's device work
        at thousands of walkers, tools/bench_vecenv_api.py), and a walker whose dict carried a terminal observation gets a fresh one at its next step -- what a caller kept from
        an earlier step is never changed under its feet.  (A caller that WRITES into the info of an unfinished walker sees its entry again at the next step; reuse_infos = False
        restores N fresh dicts per step.)"""
        n = self.num_envs
        if not self.reuse_infos:
            return [{} for _ in range(n)]
        pool = self._info_pool
        if pool is None:
            pool = self._info_pool = [{} for _ in range(n)]
        for i in self._info_dirty:
            pool[i] = {}
        self._info_dirty = []
        return list(pool)

    def rollout_fixed(self, actions, obs_out=None, rew_out=None, done_out=None):
        """Synthetic fixed-length rollout: actions float32 cuda [T, N, nu] (no policy)."""
        T = actions.shape[0]
        n, dev = self.num_envs, self.device
        obs_out = torch.empty(T, n, self.obs_dim, device=dev) if obs_out is None else obs_out
        rew_out = torch.empty(T, n, device=dev) if rew_out is None else rew_out
        done_out = torch.empty(T, n, dtype=torch.uint8, device=dev) if done_out is None else done_out
        a = actions.to(device=dev, dtype=torch.float32).contiguous()
        lib.check(self._lib.dl_rollout_fixed(self._h, T, _ptr(a), _ptr(obs_out), _ptr(rew_out), _ptr(done_out), _stream()))
        return obs_out, rew_out, done_out

    def get_attr(self, name, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        if name in MONITOR_ATTRS:
            out = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
            lib.check(self._lib.dl_stats_snapshot(self._h, name.encode(), _ptr(out), _stream()))
            if indices is None:
                return out.cpu().numpy().tolist()
            sel = out[torch.as_tensor(list(idx), dtype=torch.long, device=self.device)]      # only the requested walkers cross PCIe
            return sel.cpu().numpy().tolist()
        if name == 'ep_lens':
            return [list(self.ep_lens[i]) for i in idx]
        if name in monitor_lists.NAMES:
            if self._mlists is None:
                raise AttributeError(f'{name}: call track_monitor_lists() first (the per-episode lists of Monitor are kept on the host on request)')
            return [self._mlists.get(name, i) for i in idx]
        raise AttributeError(name)

    def set_attr(self, name, value, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        if name == 'ep_lens':
            for i in idx:
                self.ep_lens[i] = list(value)
            return
        raise AttributeError(f'cannot set {name!r} on HipVecEnv')

    def track_monitor_lists(self, on=True):
        """Keep Monitor's rsi_positions / et_positions / difficult_rsi_phases / median_abs_torque_smoothed (monitor_lists.py) from now
        on; they are updated by the numpy step_wait() surface (four device words per walker cross PCIe per control step)."""
        self._mlists = monitor_lists.MonitorLists(self.num_envs, self.cfg.ep_dur_max) if on else None

    def _update_monitor_lists(self, done_h):
        if self._mlists is None:
            return
        words = torch.empty(4, self.num_envs, dtype=torch.float64, device=self.device)
        for k, name in enumerate((b'init_pos', b'et_pos', b'last_abs_torque', b'difficult')):
            lib.check(self._lib.dl_stats_snapshot(self._h, name, _ptr(words[k]), _stream()))
        w = words.cpu().numpy()
        self._mlists.update(done_h, w[0], w[1], w[2], w[3])

    def env_is_wrapped(self, wrapper_class, indices=None):
        """SB3 VecEnv protocol: the walkers are not gym.Wrapper chains; Monitor's statistics are kept by the step kernel."""
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        # only the Monitor classes the reference wraps its envs in (drloco/common/utils.py:109: drloco.mujoco.monitor_wrapper.Monitor; SB3's own
        # Monitor is what evaluate_policy asks about) -- not any class that happens to be called Monitor
        mod = getattr(wrapper_class, '__module__', '') or ''
        is_monitor = getattr(wrapper_class, '__name__', '') == 'Monitor' and (mod.startswith('stable_baselines3.') or mod.startswith('drloco.') or mod.endswith('monitor_wrapper') or mod.endswith('.monitor'))
        return [is_monitor for _ in idx]

    def get_images(self):
        raise NotImplementedError('rendering is outside the device path (SURVEY.md 8: out of scope)')

    def render(self, mode='human'):
        raise NotImplementedError('rendering is outside the device path (SURVEY.md 8: out of scope)')

    def env_method(self, method_name, *args, indices=None, **kwargs):
        if method_name == 'activate_evaluation':
            self.activate_evaluation()
            return [None] * self.num_envs
        if method_name == 'get_walked_distance':
            return list(self.get_walked_distance())
        if method_name == 'do_terminate_early':
            return [tuple(r) for r in self.do_terminate_early()]
        raise NotImplementedError(method_name)

    # ---- MimicEnv surface used by the evaluation loop (drloco/common/callback.py:285-314) ----
    def activate_evaluation(self, on=True):
        """MimicEnv.activate_evaluation (mimic_env.py:245-249): deterministic init states from now on."""
        lib.check(self._lib.dl_set_eval(self._h, int(on)))
        self._eval = bool(on)

    def is_evaluation_on(self):
        return getattr(self, '_eval', False)

    def set_randomization(self, mass_scale=None, floor_friction=None):
        """MimicEnv.dynamics_randomization (a stub in the reference, mimic_env.py:492-524) for body masses/inertias
        (one scale per walker) and the floor friction; arrays of length N or None."""
        f = lambda a: None if a is None else torch.as_tensor(np.asarray(a, np.float32), device=self.device).contiguous()
        ms, fr = f(mass_scale), f(floor_friction)
        lib.check(self._lib.dl_set_randomization(self._h, _ptr(ms), _ptr(fr), _stream()))
        torch.cuda.current_stream().synchronize()

    def set_push(self, force=None):
        """World-frame force [N, 3] on the torso's centre of mass (xfrc_applied) for the following steps; None clears it."""
        f = None if force is None else torch.as_tensor(np.asarray(force, np.float32), device=self.device).contiguous()
        lib.check(self._lib.dl_set_push(self._h, _ptr(f), _stream()))
        torch.cuda.current_stream().synchronize()

    def set_split(self, on=True):
        """dl_set_split: the eight-wave workgroup form of the step kernel (dynamics waves + look-ahead partner waves; both walkers, float32,
        16 lanes per walker).  Shorter launches, but nothing else runs next to it: for single-handle use (bench.py switches it on)."""
        lib.check(self._lib.dl_set_split(self._h, int(bool(on))))
        self.split = bool(on)

    def set_push_schedule(self, force=None, phase=None, period=400, duration=20):
        """Periodic pushes kept on the device: walker w is pushed with force[w] during the k-th control step from now iff
        (k + phase[w]) % period < duration (defaults: 0.1 s every 2 s at 200 Hz).  force None switches the schedule off."""
        f = None if force is None else torch.as_tensor(np.asarray(force, np.float32), device=self.device).contiguous()
        ph = None if phase is None else torch.as_tensor(np.asarray(phase, np.int32), device=self.device).contiguous()
        lib.check(self._lib.dl_set_push_schedule(self._h, _ptr(f), _ptr(ph), int(period), int(duration), _stream()))
        torch.cuda.current_stream().synchronize()

    def do_terminate_early(self):
        """MimicEnv.do_terminate_early (mimic_env.py:652-702) for all walkers: bool [N, 4] =
        (terminate_early, com_height_too_low, trunk_ang_exceeded, is_drunk)."""
        flags = torch.empty(self.num_envs, 4, dtype=torch.int32, device=self.device)
        lib.check(self._lib.dl_terminate_early(self._h, _ptr(flags), _stream()))
        return flags.cpu().numpy().astype(bool)

    def get_walked_distance(self):
        walked = torch.empty(self.num_envs, dtype=torch.float64, device=self.device)
        lib.check(self._lib.dl_get_state(self._h, None, None, None, None, _ptr(walked), _stream()))
        return walked.cpu().numpy()

    @property
    def envs(self):
        """`eval_env.venv.envs[0].env` of the reference's evaluation loop resolves to per-walker views."""
        return [_WalkerView(self, i) for i in range(self.num_envs)]

    def seed(self, seed=None):
        # the reference seeds only gym's np_random, which the env never uses (utils.py:113);
        # RSI draws come from the counter-based stream keyed by cfg.seed
        return [seed] * self.num_envs

    def close(self):
        if getattr(self, '_h', None):
            self._lib.dl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parity hooks ---------------------------------------------------------------------
    def get_state(self):
        n, dev = self.num_envs, self.device
        q = torch.empty(self.nv, n, dtype=self.rdtype, device=dev); v = torch.empty_like(q); w = torch.empty_like(q)
        cur = torch.empty(abi.DL_CUR_WORDS, n, dtype=torch.int32, device=dev)
        walked = torch.empty(n, dtype=torch.float64, device=dev)
        lib.check(self._lib.dl_get_state(self._h, _ptr(q), _ptr(v), _ptr(w), _ptr(cur), _ptr(walked), _stream()))
        return dict(qpos=q.cpu().numpy(), qvel=v.cpu().numpy(), warm=w.cpu().numpy(), cursor=cur.cpu().numpy(), walked=walked.cpu().numpy())

    def set_state(self, qpos=None, qvel=None, warm=None, cursor=None, walked=None):
        dev = self.device
        f = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev).contiguous()
        q, v, w = f(qpos, self.rdtype), f(qvel, self.rdtype), f(warm, self.rdtype)
        c, wk = f(cursor, torch.int32), f(walked, torch.float64)
        lib.check(self._lib.dl_set_state(self._h, _ptr(q), _ptr(v), _ptr(w), _ptr(c), _ptr(wk), _stream()))
        torch.cuda.current_stream().synchronize()

    def get_ref_offsets(self):
        """Quirk Q4's record (dl_get_ref_offsets): the COM-z offset every reference step of every walker's data set carries from the last reset that landed on it
        (adjust_COM_Z_pos, base_ref_trajecs.py:126-127): numpy [n_steps, N]."""
        z = torch.empty(self.refs.n_steps, self.num_envs, dtype=self.rdtype, device=self.device)
        lib.check(self._lib.dl_get_ref_offsets(self._h, _ptr(z), _stream()))
        return z.cpu().numpy()

    def set_ref_offsets(self, z):
        z = torch.as_tensor(np.ascontiguousarray(z), dtype=self.rdtype, device=self.device).contiguous()
        if tuple(z.shape) != (self.refs.n_steps, self.num_envs):
            raise ValueError(f'z_offsets must be [n_steps = {self.refs.n_steps}, N = {self.num_envs}]')
        lib.check(self._lib.dl_set_ref_offsets(self._h, _ptr(z), _stream()))
        torch.cuda.current_stream().synchronize()

    def forward(self, ctrl=None):
        n, dev = self.num_envs, self.device
        u = None if ctrl is None else torch.as_tensor(np.ascontiguousarray(ctrl), dtype=self.rdtype, device=dev)
        qacc = torch.empty(self.nv, n, dtype=self.rdtype, device=dev)
        ncon = torch.empty(n, dtype=torch.int32, device=dev); nefc = torch.empty_like(ncon); nit = torch.empty_like(ncon)
        lib.check(self._lib.dl_forward(self._h, _ptr(u), _ptr(qacc), _ptr(ncon), _ptr(nefc), _ptr(nit), _stream()))
        return qacc.cpu().numpy(), ncon.cpu().numpy(), nefc.cpu().numpy(), nit.cpu().numpy()

    def debug_counters(self, clear=True):
        """Solver diagnostics of the 16-lane step kernel: int32 [4, N] = sum iterations, max iterations of the
        last step, sum constraint rows, diverged steps (since the last clear).  The first call enables them."""
        out = torch.zeros(4, self.num_envs, dtype=torch.int32, device=self.device)
        lib.check(self._lib.dl_debug_counters(self._h, _ptr(out), int(clear), _stream()))
        return out.cpu().numpy()

    def debug_capstate(self):
        out = torch.zeros(48, self.num_envs, device=self.device)
        lib.check(self._lib.dl_debug_capstate(self._h, _ptr(out), _stream()))
        return out.cpu().numpy()

    def debug_eval_iters(self, rows=False):
        """Newton iterations of every walker in the 4 x frame_skip forward evaluations of the last control step: int [evals, N]
        (16-lane kernels, after debug_counters() has enabled the diagnostics; tools/diag_lockstep.py).  rows=True: (iterations, constraint rows)."""
        out = torch.zeros(40, self.num_envs, device=self.device)
        lib.check(self._lib.dl_debug_eval_iters(self._h, _ptr(out), _stream()))
        v = out[:4 * self.model.frame_skip].cpu().numpy().astype(np.int32)
        return (v % 128, v // 128) if rows else v % 128

    def debug_last_ctrl(self):
        """sim.data.ctrl of the last step() (after _rescale_actions / mirror_action), float32 [N, nu]; the first call enables
        the record (and returns zeros)."""
        out = torch.zeros(self.num_envs, self.nu, device=self.device)
        lib.check(self._lib.dl_debug_last_ctrl(self._h, _ptr(out), _stream()))
        return out.cpu().numpy()

    def debug_inject(self, qpos=None, qvel=None, flags=None, rsi=None):
        dev = self.device
        f = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev).contiguous()
        q, v, fl, r = f(qpos, self.rdtype), f(qvel, self.rdtype), f(flags, torch.int32), f(rsi, torch.int32)
        lib.check(self._lib.dl_debug_inject(self._h, _ptr(q), _ptr(v), _ptr(fl), _ptr(r), _stream()))
        torch.cuda.current_stream().synchronize()


class _WalkerView:
    """Stands in for `Monitor(MimicEnv)` of one walker: `.env` is the walker itself."""

    def __init__(self, venv, index):
        self._venv, self._i = venv, index

    @property
    def env(self):
        return self

    def activate_evaluation(self):
        self._venv.activate_evaluation()        # a handle-wide switch (the reference evaluates on a 1-env DummyVecEnv)

    def is_evaluation_on(self):
        return self._venv.is_evaluation_on()

    def get_walked_distance(self):
        return float(self._venv.get_walked_distance()[self._i])

    def __getattr__(self, name):
        if name in MONITOR_ATTRS or name == 'ep_lens':
            return self._venv.get_attr(name, self._i)[0]
        raise AttributeError(name)


def merge_moments_across_ranks(mean, var, count, mean0, var0, count0, group=None):
    """Exact cross-rank merge of running moments (collective C3).  Every rank holds
    (mean, var, count) = the state agreed at the last merge (mean0, var0, count0) advanced by its OWN batches.  The
    increments [n, sum, sum of squares] since that state are all-reduced and added to it, which gives the moments a
    single process would have after seeing every rank's batches (one all-reduce of 2 D + 1 doubles).  Tensors are
    float64 (count: 1 element) on any device; updated in place, returns nothing."""
    import torch.distributed as dist
    n0, n1 = count0, count
    inc = torch.cat([(n1 * mean - n0 * mean0).reshape(-1), (n1 * (var + mean * mean) - n0 * (var0 + mean0 * mean0)).reshape(-1), (n1 - n0).reshape(-1)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(inc, group=group)
    d = mean.numel()
    n = n0 + inc[2 * d]
    s = n0 * mean0 + inc[:d].reshape(mean.shape)
    q = n0 * (var0 + mean0 * mean0) + inc[d:2 * d].reshape(mean.shape)
    new_mean = s / n
    mean.copy_(new_mean)
    var.copy_(torch.clamp(q / n - new_mean * new_mean, min=0))
    count.copy_(n.reshape(count.shape))
    mean0.copy_(mean); var0.copy_(var); count0.copy_(count)


class RunningMeanStd:
    """SB3 1.0 RunningMeanStd with device-resident float64 moments."""

    def __init__(self, shape, device, epsilon=1e-4):
        d = int(np.prod(shape)) if shape else 1
        self.shape = tuple(shape)
        self._mean = torch.zeros(d, dtype=torch.float64, device=device)
        self._var = torch.ones(d, dtype=torch.float64, device=device)
        self._count = torch.full((1,), epsilon, dtype=torch.float64, device=device)
        self._sync = (self._mean.clone(), self._var.clone(), self._count.clone())     # state at the last cross-rank merge

    def sync(self, group=None):
        merge_moments_across_ranks(self._mean, self._var, self._count, *self._sync, group=group)

    @property
    def mean(self):
        return self._mean.cpu().numpy().reshape(self.shape)

    @property
    def var(self):
        return self._var.cpu().numpy().reshape(self.shape)

    @property
    def count(self):
        return float(self._count.item())

    def state(self):
        return dict(mean=self.mean, var=self.var, count=self.count)

    def load_state(self, s):
        self._mean.copy_(torch.as_tensor(np.asarray(s['mean'], np.float64).reshape(-1)))
        self._var.copy_(torch.as_tensor(np.asarray(s['var'], np.float64).reshape(-1)))
        self._count.fill_(float(s['count']))
        self._sync = (self._mean.clone(), self._var.clone(), self._count.clone())


class HipVecNormalize(_VecEnvWrapperBase):
    """VecNormalize(venv, norm_obs=True, norm_reward=True, clip 10, gamma 0.99, eps 1e-8) on device.

    Two points where SB3 releases differ are switches (SB3 1.0's source is not available here, so the
    defaults follow what is known of 1.0 and the alternatives can be selected when a checkpoint says otherwise):
      reset_moments          what reset() feeds into the running moments while training: 'none' (default), 'ret' (the zeroed
                             returns into ret_rms, as the baselines-derived releases up to 1.0 may do) or 'obs' (the first
                             observations into obs_rms, as later releases do);
      norm_terminal_obs      whether infos['terminal_observation'] is normalised (later releases) or raw (1.0, default)."""

    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0,
                 gamma=0.99, epsilon=1e-8, reset_moments='none', norm_terminal_obs=False, sync='per_rollout', process_group=None):
        if reset_moments not in ('none', 'ret', 'obs'):
            raise ValueError("reset_moments must be 'none', 'ret' or 'obs'")
        if sync not in ('per_rollout', 'per_step'):
            raise ValueError("sync must be 'per_rollout' or 'per_step'")
        # data-parallel runs (one process per GPU): 'per_rollout' advances the moments per rank and merges them exactly between rollouts
        # (sync_moments(): no collective inside a rollout); 'per_step' is SB3's semantics across ranks -- every control step's update uses the
        # batch of ALL ranks: one all-reduce of 2 (obs_dim + 1) doubles per control step (step_tensors / step; not the long fixed-action launches)
        self.sync, self.process_group = sync, process_group
        _VecEnvWrapperBase.__init__(self, venv)          # venv, num_envs, observation_space, action_space
        self.reset_moments, self.norm_terminal_obs = reset_moments, norm_terminal_obs
        self._lib = venv._lib
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        dev = venv.device
        self.obs_rms = RunningMeanStd((venv.obs_dim,), dev)
        self.ret_rms = RunningMeanStd((), dev)
        self.ret = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)
        self.norm_obs_t = torch.zeros_like(venv.obs)
        self.norm_rew_t = torch.zeros_like(venv.rew)
        self._vn_work = torch.zeros(abi.vn_workspace_bytes(venv.obs_dim) // 8, dtype=torch.float64, device=dev)    # DL_VN_WORKSPACE_BYTES
        self._ov = None                                   # overlap state (enable_overlap)
        # form of the moment reduction (flags bit 16 of dl_vecnormalize_step): one workgroup (lowest latency when the next launch waits
        # for it: a policy in the loop with <= 4096 walkers -- 14 us) or 32 blocks (keeps its pace on a side stream under a running
        # env-step kernel: enable_overlap; and for large batches, where one CU's time grows with the batch: 236 us for 16 384 walkers
        # under contention).  Both are deterministic; they sum in different orders, i.e. their moments can differ in the last bit.
        self.multi_block_reduce = self.num_envs > 4096
        # flags bit 32: the "blocked" summation order (blocks of 16 rows -> <= 8 groups -> total), which the persistent rollout kernel of
        # dl_collect_rollouts follows by construction: with it a launch-per-step rollout and a persistent one agree bit for bit
        self.blocked_reduce = False
        # fixed-action runs (steps_fixed): normalise the K steps of a run with dl_vecnormalize_steps (six small launches) instead of K x
        # dl_vecnormalize_step; the moments then agree with the step-by-step form to rounding (the shift of the sums differs), not bit for bit
        self.batched_steps = False

    # the raw outputs of the last step stay in the env's own tensors (get_original_obs / get_original_reward)
    @property
    def old_obs(self):
        if self._ov is not None and self._ov['last'] is not None:
            self.flush()
            return self._ov['raw'][self._ov['last']][0]
        return self.venv.obs

    @property
    def old_rew(self):
        if self._ov is not None and self._ov['last'] is not None:
            self.flush()
            return self._ov['raw'][self._ov['last']][1]
        return self.venv.rew

    def _normalize_obs_inplace(self, x):
        n = x.shape[0]
        lib.check(self._lib.dl_normalize_obs(_ptr(x), _ptr(self.obs_rms._mean), _ptr(self.obs_rms._var), n, x.shape[1],
                                             self.epsilon, self.clip_obs, _stream()))

    def _flags(self):
        # SB3 1.0 step_wait: obs_rms / ret_rms (and ret) advance whenever training is on, whatever norm_obs / norm_reward say
        # (the reference's load_env builds an evaluation env with norm_reward=False); the norm_* switches only gate the scaling
        return (1 if self.training else 0) | (2 if self.norm_obs else 0) | (4 if self.training else 0) | (8 if self.norm_reward else 0) | \
               (32 if self.blocked_reduce else (16 if self.multi_block_reduce else 0))

    def state_struct(self):
        """dl_vecnorm_state for dl_rollout_policy: pointers to the device-resident moments of this object."""
        st = abi.VecNormState()
        st.obs_mean, st.obs_var, st.obs_count = self.obs_rms._mean.data_ptr(), self.obs_rms._var.data_ptr(), self.obs_rms._count.data_ptr()
        st.ret, st.ret_mean, st.ret_var, st.ret_count = self.ret.data_ptr(), self.ret_rms._mean.data_ptr(), self.ret_rms._var.data_ptr(), self.ret_rms._count.data_ptr()
        st.workspace = self._vn_work.data_ptr()
        st.gamma, st.eps, st.clip_obs, st.clip_rew, st.flags = self.gamma, self.epsilon, self.clip_obs, self.clip_reward, self._flags()
        return st

    def _vn_launch(self, obs, rew, done, obs_out, rew_out):
        n, d = obs.shape
        flags = self._flags()
        if self.sync == 'per_step' and self.training:
            import torch.distributed as dist
            world = dist.get_world_size(self.process_group) if (dist.is_available() and dist.is_initialized()) else 1
            if getattr(self, '_sums', None) is None:
                self._sums = torch.zeros(2 * (d + 1), dtype=torch.float64, device=obs.device)
            lib.check(self._lib.dl_vn_local_sums(_ptr(obs), _ptr(rew), _ptr(self.obs_rms._mean), _ptr(self.ret), _ptr(self.ret_rms._mean), n, d, self.gamma, flags,
                                                 _ptr(self._sums), _stream()))
            if world > 1:
                if dist.get_backend(self.process_group) == 'gloo':          # (CPU collectives: the one-GPU test rig)
                    h = self._sums.cpu()
                    dist.all_reduce(h, group=self.process_group)
                    self._sums.copy_(h)
                else:
                    dist.all_reduce(self._sums, group=self.process_group)
            lib.check(self._lib.dl_vn_merge_sums(_ptr(self._sums), n * world, _ptr(self.obs_rms._mean), _ptr(self.obs_rms._var), _ptr(self.obs_rms._count),
                                                 _ptr(self.ret_rms._mean), _ptr(self.ret_rms._var), _ptr(self.ret_rms._count), d, flags, _stream()))
            flags |= 64
        lib.check(self._lib.dl_vecnormalize_step(
            _ptr(obs), _ptr(rew), _ptr(done), _ptr(self.obs_rms._mean), _ptr(self.obs_rms._var), _ptr(self.obs_rms._count),
            _ptr(self.ret), _ptr(self.ret_rms._mean), _ptr(self.ret_rms._var), _ptr(self.ret_rms._count), n, d,
            self.gamma, self.epsilon, self.clip_obs, self.clip_reward, flags, _ptr(obs_out), _ptr(rew_out), _ptr(self._vn_work), _stream()))

    def step_tensors(self, actions, obs_out=None, rew_out=None, done_out=None):
        """One control step + VecNormalize.step_wait, everything on the device: dl_step, then the two launches
        of dl_vecnormalize_step.  obs_out / rew_out / done_out may be rollout-buffer slots (float32 [N, obs],
        float32 [N], uint8 [N]); by default the results land in norm_obs_t / norm_rew_t / venv.done.
        With `enable_overlap()` the normalisation of step t runs on a side stream while the main stream already
        simulates step t + 1 (see there)."""
        obs_out = self.norm_obs_t if obs_out is None else obs_out
        rew_out = self.norm_rew_t if rew_out is None else rew_out
        if self._ov is None:
            obs, rew, done, term = self.venv.step_tensors(actions, done_out=done_out)
            self._vn_launch(obs, rew, done, obs_out, rew_out)
            return obs_out, rew_out, done, term
        ov = self._ov
        C = ov['chunk']
        k = ov['k']                                          # slot in the ring of 2 C raw-buffer sets
        main = torch.cuda.current_stream()
        if k % C == 0 and ov['read'][k // C] is not None:
            main.wait_event(ov['read'][k // C])             # the normalisations that last read this half of the ring
        raw_obs, raw_rew = ov['raw'][k]
        done = ov['done'][k] if done_out is None else done_out
        _, _, done, term = self.venv.step_tensors(actions, done_out=done, obs_out=raw_obs, rew_out=raw_rew)
        ov['pending'].append((k, done, obs_out, rew_out))
        ov['last'] = k
        ov['k'] = (k + 1) % (2 * C)
        if (k + 1) % C == 0:
            self._submit_pending()
        return obs_out, rew_out, done, term

    def _submit_pending(self):
        """Hand the steps simulated since the last hand-over to the side stream: ONE event pair per chunk (event packets
        between the env-step kernels cost launch gap)."""
        ov = self._ov
        if not ov['pending']:
            return
        main, side = torch.cuda.current_stream(), ov['stream']
        half = ov['pending'][0][0] // ov['chunk']
        ov['stepped'][half].record(main)
        side.wait_event(ov['stepped'][half])
        pending, ov['pending'] = ov['pending'], []         # cleared whatever happens below: a failed hand-over must not be re-issued
        with torch.cuda.stream(side):
            # the batched form needs consecutive ring slots and one contiguous block of done rows (steps_fixed produces exactly that;
            # step_tensors callers that pass their own done_out slots may not): otherwise one dl_vecnormalize_step per step
            if self.batched_steps and len(pending) > 1 and self._run_is_contiguous(pending):
                self._vn_launch_steps(pending)
            else:
                for k, done, obs_out, rew_out in pending:
                    self._vn_launch(ov['raw'][k][0], ov['raw'][k][1], done, obs_out, rew_out)
            ov['read'][half] = ov['readev'][half]
            ov['read'][half].record(side)

    def _run_is_contiguous(self, pending):
        n, k0 = self.venv.num_envs, pending[0][0]
        return all(p[0] == k0 + i and p[1].is_contiguous() and p[1].data_ptr() == pending[0][1].data_ptr() + i * n for i, p in enumerate(pending))

    def _vn_launch_steps(self, pending):
        """dl_vecnormalize_steps for a run of consecutive ring slots: six small launches instead of two per control step (the moments agree
        with the step-by-step form to rounding, include/drloco_hip.h).  pending: [(ring slot, done row, obs_out, rew_out)], slots consecutive."""
        ov = self._ov
        K, k0 = len(pending), pending[0][0]
        n, d = self.venv.obs.shape
        assert self._run_is_contiguous(pending)
        need = abi.vn_steps_workspace_bytes(K, n, d)
        if ov.get('steps_work') is None or ov['steps_work'].numel() < need:
            ov['steps_work'] = torch.empty(need, dtype=torch.uint8, device=self.venv.device)
        key = (k0, tuple(p[2].data_ptr() for p in pending), tuple(p[3].data_ptr() for p in pending))
        ptrs = ov.setdefault('steps_ptrs', {})
        if key not in ptrs:         # destinations of a rollout repeat from rollout to rollout: one upload per distinct run
            if len(ptrs) > 16:
                ptrs.clear()
            ptrs[key] = (torch.tensor(key[1], dtype=torch.int64).to(self.venv.device), torch.tensor(key[2], dtype=torch.int64).to(self.venv.device))
        po, pr = ptrs[key]
        st = self.state_struct()
        lib.check(self._lib.dl_vecnormalize_steps(C.byref(st), K, _ptr(ov['raw_obs'][k0]), _ptr(ov['raw_rew'][k0]), _ptr(pending[0][1]), n, d, _ptr(po), _ptr(pr),
                                                  _ptr(ov['steps_work']), _stream()))

    def enable_overlap(self, chunk=8):
        """Software pipelining for callers whose next actions do not depend on this step's normalised observation
        (fixed-action rollouts such as BASELINE configs[1]; NOT a policy in the loop): dl_step writes its raw
        observation / reward into a ring of 2 x `chunk` buffer sets, and after every `chunk` steps their
        dl_vecnormalize_step launches are issued on a side HIP stream, where they run while the main stream keeps
        simulating.  Same kernels, same order of moment updates, same results; the outputs of step_tensors are complete
        only after `flush()` (call it before anything on the main stream reads them: GAE, the policy, a copy to the host)."""
        dev = self.venv.device
        self.multi_block_reduce = True
        if getattr(self.venv, 'split', False):
            # nothing hides under split workgroups (they fill the GPU): the normalisations of a run go through dl_vecnormalize_steps (six small
            # launches per run; moments agree with the step-by-step form to rounding, not bit for bit).  Set batched_steps = False to opt out.
            self.batched_steps = True
        ev = lambda: torch.cuda.Event()
        # high priority: its own hardware-queue pool (a default-priority stream may share the main stream's queue once other
        # libraries -- RCCL -- have created streams) and the tiny launches are dispatched while the step kernel runs
        n, d = self.venv.obs.shape
        raw_obs, raw_rew = torch.zeros(2 * chunk, n, d, device=dev), torch.zeros(2 * chunk, n, device=dev)     # contiguous: dl_rollout_fixed writes runs of slots
        done = torch.zeros(2 * chunk, n, dtype=torch.uint8, device=dev)
        self._ov = dict(stream=torch.cuda.Stream(device=dev, priority=-1), chunk=int(chunk), k=0, last=None, pending=[], read=[None, None], readev=[ev(), ev()],
                        stepped=[ev(), ev()], raw_obs=raw_obs, raw_rew=raw_rew,
                        raw=[(raw_obs[i], raw_rew[i]) for i in range(2 * chunk)], done=[done[i] for i in range(2 * chunk)])

    def steps_fixed(self, actions, obs_outs, rew_outs, done_out):
        """K <= chunk control steps with a pre-generated action tape (float32 [K, N, nu], contiguous) in ONE dl_rollout_fixed
        call -- the 16-lane kernel then takes all K steps in one launch (K <= 512) -- followed by the K normalisations on the side
        stream.  done_out: uint8 [K, N] contiguous (e.g. rows t+1.. of the episode-start array); obs_outs / rew_outs: K
        destinations each (rollout-buffer slots).  Needs enable_overlap(); results are complete after flush()."""
        ov = self._ov
        assert ov is not None, 'enable_overlap() first'
        C, K = ov['chunk'], actions.shape[0]
        assert 1 <= K <= C and len(obs_outs) == K and len(rew_outs) == K and done_out.shape[0] == K and done_out.is_contiguous()
        if ov['pending']:
            self._submit_pending()
        k0 = ((ov['k'] + C - 1) // C * C) % (2 * C)          # start of a ring half
        half = k0 // C
        main = torch.cuda.current_stream()
        if ov['read'][half] is not None:
            main.wait_event(ov['read'][half])
        self.venv.rollout_fixed(actions, obs_out=ov['raw_obs'][k0:k0 + K], rew_out=ov['raw_rew'][k0:k0 + K], done_out=done_out)
        ov['pending'] = [(k0 + i, done_out[i], obs_outs[i], rew_outs[i]) for i in range(K)]
        ov['last'] = k0 + K - 1
        ov['k'] = (k0 + C) % (2 * C)
        self._submit_pending()

    def flush(self):
        """Issue what is still pending and make the main stream wait for the side stream."""
        if self._ov is not None:
            if self._ov['pending']:
                # a partial chunk: continue with the other half of the ring afterwards
                C, k = self._ov['chunk'], self._ov['k']
                self._submit_pending()
                self._ov['k'] = ((k + C - 1) // C * C) % (2 * C)
            main = torch.cuda.current_stream()
            for e in self._ov['read']:
                if e is not None:
                    main.wait_event(e)

    def sync_moments(self, process_group=None):
        """Data-parallel runs: make the observation / return moments of all ranks those of the union of their batches
        (call between rollouts; every rank then normalises identically, as the single-process reference does)."""
        if self.sync == 'per_step':
            for r in (self.obs_rms, self.ret_rms):          # nothing to merge: every step's update already used all ranks' batches
                r._sync = (r._mean.clone(), r._var.clone(), r._count.clone())
            return
        self.obs_rms.sync(process_group)
        self.ret_rms.sync(process_group)

    def step_async(self, actions):
        self._actions = torch.as_tensor(np.asarray(actions), dtype=torch.float32, device=self.venv.device)

    def step_wait(self):
        obs, rew, done, term = self.step_tensors(self._actions)
        self.flush()
        done_h = done.cpu().numpy().astype(bool)
        infos = self.venv._infos()
        self.venv._ep_len += 1
        self.venv._update_monitor_lists(done_h)
        if done_h.any():
            t = term.clone()
            if self.norm_obs and self.norm_terminal_obs:
                self._normalize_obs_inplace(t)
            term_h = t.cpu().numpy()
            for i in np.nonzero(done_h)[0]:
                infos[i]['terminal_observation'] = term_h[i]
                self.venv._info_dirty.append(int(i))
                self.venv.ep_lens[i].append(int(self.venv._ep_len[i]))
                self.venv._ep_len[i] = 0
        return obs.cpu().numpy(), rew.cpu().numpy(), done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def reset(self):
        """ret = 0; the first observation is normalised; what enters the moments is the reset_moments switch (see the class)."""
        self.venv.reset_tensors()
        self.ret.zero_()
        if self.training and self.reset_moments == 'ret':
            z = torch.zeros(self.num_envs, 1, device=self.venv.device)
            lib.check(self._lib.dl_moments_update(_ptr(self.ret_rms._mean), _ptr(self.ret_rms._var), _ptr(self.ret_rms._count), _ptr(z), self.num_envs, 1, _stream()))
        elif self.training and self.reset_moments == 'obs':
            lib.check(self._lib.dl_moments_update(_ptr(self.obs_rms._mean), _ptr(self.obs_rms._var), _ptr(self.obs_rms._count), _ptr(self.venv.obs), self.num_envs,
                                                  self.venv.obs_dim, _stream()))
        self.norm_obs_t.copy_(self.venv.obs)
        if self.norm_obs:
            self._normalize_obs_inplace(self.norm_obs_t)
        return self.norm_obs_t.cpu().numpy()

    def normalize_obs(self, obs):
        x = torch.as_tensor(np.asarray(obs, np.float32), device=self.venv.device).reshape(-1, self.venv.obs_dim).clone()
        if self.norm_obs:
            self._normalize_obs_inplace(x)
        return x.cpu().numpy().reshape(np.shape(obs))

    def get_original_obs(self):
        return self.old_obs.cpu().numpy()

    def get_original_reward(self):
        return self.old_rew.cpu().numpy()

    def get_attr(self, name, indices=None):
        return self.venv.get_attr(name, indices)

    def set_attr(self, name, value, indices=None):
        return self.venv.set_attr(name, value, indices)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return self.venv.env_method(method_name, *args, indices=indices, **kwargs)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return self.venv.env_is_wrapped(wrapper_class, indices)

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def close(self):
        self.venv.close()

    # utils.save_model / load_env (drloco/common/utils.py:175-192,234-240) keep the running moments
    def save(self, path, sb3_format=False):
        """sb3_format: write SB3 1.0's object pickle (drloco_amd.checkpoint) instead of this package's plain dict."""
        if sb3_format:
            from .checkpoint import write_vecnormalize_sb3
            return write_vecnormalize_sb3(self, path)
        with open(path, 'wb') as f:
            pickle.dump(dict(obs_rms=self.obs_rms.state(), ret_rms=self.ret_rms.state(), clip_obs=self.clip_obs,
                             clip_reward=self.clip_reward, gamma=self.gamma, epsilon=self.epsilon,
                             norm_obs=self.norm_obs, norm_reward=self.norm_reward, training=self.training), f)

    @staticmethod
    def load(path, venv):
        """Reads this package's dict pickle or an SB3 1.0 `VecNormalize.save` file (e.g. one written by the reference)."""
        from .checkpoint import read_vecnormalize
        s = read_vecnormalize(path)
        vn = HipVecNormalize(venv, training=bool(s.get('training', True)), norm_obs=s['norm_obs'], norm_reward=s['norm_reward'], clip_obs=s['clip_obs'],
                             clip_reward=s['clip_reward'], gamma=s['gamma'], epsilon=s['epsilon'])
        vn.obs_rms.load_state(s['obs_rms'])
        vn.ret_rms.load_state(s['ret_rms'])
        return vn


def vec_env(env_id=models.STRAIGHT_WALKER, num_envs=4096, seed=33, norm_rew=True, load_path=None, **kw):
    """Drop-in for drloco.common.utils.vec_env (utils.py:97-134)."""
    venv = HipVecEnv(env_id, num_envs=num_envs, seed=seed, **kw)
    if load_path is not None:
        return HipVecNormalize.load(load_path, venv)
    return HipVecNormalize(venv, norm_obs=True, norm_reward=norm_rew)
