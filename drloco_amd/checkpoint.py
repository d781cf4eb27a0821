"""Checkpoint formats either side of the path: what `utils.save_model` writes and
`utils.load_env` / `PPO.load` read (drloco/common/utils.py:175-192,234-240; callback.py:280-291).

  envs/env_<ckpt>       [3P SB3 1.0] `VecNormalize.save`: `pickle.dump(self)` with `__getstate__` dropping venv,
                        class_attributes and ret -- an object of class
                        stable_baselines3.common.vec_env.vec_normalize.VecNormalize whose obs_rms / ret_rms are
                        stable_baselines3.common.running_mean_std.RunningMeanStd(mean, var, count) and whose
                        observation_space / action_space are gym Boxes.
  models/model_<ckpt>.zip  [3P SB3 1.0] zip archive: `data` (json of the constructor arguments), `policy.pth`
                        (`th.save(policy.state_dict())`), `policy.optimizer.pth`, `pytorch_variables.pth`,
                        `_stable_baselines3_version`.  For the reference's CustomActorCriticPolicy
                        (drloco/custom/policies.py:13-80) the state dict holds log_std, mlp_extractor.policy_net.{0,2}.*
                        (the SAME modules appear again as mlp_extractor.value_net.{0,2}.*), action_net.*, value_net.*.

Neither SB3 nor gym is installed here, so both layouts are restated from SB3 1.0's published source and are
**unpinned by a reference artefact**: the reader below is tolerant (any class it cannot import becomes a plain attribute
bag), the writer emits the SB3 class paths so that an SB3 installation unpickles real objects.

What is interchangeable with the reference: `policy.pth` (the state dict, both directions) and the VecNormalize statistics
(`obs_rms`, `ret_rms`, clip values, gamma, epsilon, the norm_* and training flags).  `write_policy_zip` writes the weights only;
`write_model_zip` writes the archive in the shape of SB3 1.0's `save_to_zip_file` (stable_baselines3/common/save_util.py): `data`
as `data_to_json` leaves it (JSON values as they are, everything else as {":type:", ":serialized:" = base64(cloudpickle)} -- the
observation / action spaces, the policy class, the schedules), `policy.pth`, `policy.optimizer.pth` (Adam's state dict over the nine
parameters in `ActorCriticPolicy.parameters()` order), `pytorch_variables.pth`, `_stable_baselines3_version`.  Checked here by a round trip
through a `load_from_zip_file`-shaped reader restated in tests/stubs (tests/test_interop.py); it stays UNPINNED until an archive written by a
real SB3 1.0 exists to compare with.
"""
import io
import json
import os
import pickle
import sys
import types
import zipfile

import numpy as np
import torch

from . import abi

_VN_CLASS = ('stable_baselines3.common.vec_env.vec_normalize', 'VecNormalize')
_RMS_CLASS = ('stable_baselines3.common.running_mean_std', 'RunningMeanStd')


class _Bag:
    """Stand-in for a class that cannot be imported while reading a pickle."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):      # (dict, slots)
            state = {**(state[0] or {}), **state[1]}
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__['_state'] = state


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (_Bag,), {'__module__': module})


def read_vecnormalize(path):
    """-> dict(obs_rms, ret_rms, clip_obs, clip_reward, gamma, epsilon, norm_obs, norm_reward, training) from either
    the SB3 object pickle or this package's own dict pickle (HipVecNormalize.save)."""
    with open(path, 'rb') as f:
        obj = _TolerantUnpickler(f).load()
    if isinstance(obj, dict):
        return obj
    d = obj.__dict__
    rms = lambda r: dict(mean=np.asarray(r.mean, np.float64), var=np.asarray(r.var, np.float64), count=float(r.count))
    return dict(obs_rms=rms(d['obs_rms']), ret_rms=rms(d['ret_rms']), clip_obs=float(d['clip_obs']), clip_reward=float(d['clip_reward']),
                gamma=float(d['gamma']), epsilon=float(d['epsilon']), norm_obs=bool(d.get('norm_obs', True)),
                norm_reward=bool(d.get('norm_reward', True)), training=bool(d.get('training', True)))


def _sb3_class(module, name):
    """The real class when SB3 is importable, otherwise a placeholder registered under the SB3 module path so that
    pickle writes `module.name` as the global."""
    try:
        __import__(module)
        return getattr(sys.modules[module], name), False
    except Exception:
        parts = module.split('.')
        for i in range(1, len(parts) + 1):
            sys.modules.setdefault('.'.join(parts[:i]), types.ModuleType('.'.join(parts[:i])))
        cls = type(name, (object,), {'__module__': module})
        setattr(sys.modules[module], name, cls)
        return cls, True


def _drop_placeholders():
    for m in [m for m in sys.modules if m.split('.')[0] == 'stable_baselines3' and not hasattr(sys.modules[m], '__file__')]:
        del sys.modules[m]


def write_vecnormalize_sb3(vn, path):
    """HipVecNormalize -> the SB3 1.0 object pickle (state after `__getstate__`; the spaces are left to `set_venv`,
    which takes them from the wrapped venv on load)."""
    vn_cls, fake1 = _sb3_class(*_VN_CLASS)
    rms_cls, fake2 = _sb3_class(*_RMS_CLASS)

    def rms(r):
        o = rms_cls.__new__(rms_cls)
        o.__dict__.update(mean=np.array(r.mean, np.float64), var=np.array(r.var, np.float64), count=float(r.count))
        return o
    o = vn_cls.__new__(vn_cls)
    o.__dict__.update(num_envs=vn.num_envs, obs_rms=rms(vn.obs_rms), ret_rms=rms(vn.ret_rms), clip_obs=vn.clip_obs, clip_reward=vn.clip_reward,
                      gamma=vn.gamma, epsilon=vn.epsilon, training=vn.training, norm_obs=vn.norm_obs, norm_reward=vn.norm_reward,
                      old_obs=np.array([]), old_reward=np.array([]))
    try:
        with open(path, 'wb') as f:
            pickle.dump(o, f)
    finally:
        if fake1 or fake2:
            _drop_placeholders()


# ---- model.zip ---------------------------------------------------------------------------------
_KEYS = dict(w1='mlp_extractor.policy_net.0.weight', b1='mlp_extractor.policy_net.0.bias', w2='mlp_extractor.policy_net.2.weight',
             b2='mlp_extractor.policy_net.2.bias', wa='action_net.weight', ba='action_net.bias', wv='value_net.weight', bv='value_net.bias',
             log_std='log_std')


def read_policy_zip(path):
    """SB3 model.zip -> the nine tensors HipPolicy.load_state takes.  Only the reference's shared-trunk
    CustomActorCriticPolicy maps onto the fused kernel; separate pi/vf trunks (SB3's MlpPolicy with
    net_arch=[dict(pi=..., vf=...)], the reference's non-default branch, train.py:106-108) are refused."""
    with zipfile.ZipFile(path) as z:
        sd = torch.load(io.BytesIO(z.read('policy.pth')), map_location='cpu', weights_only=True)
    for a, b in (('mlp_extractor.policy_net.0.weight', 'mlp_extractor.value_net.0.weight'), ('mlp_extractor.policy_net.2.weight', 'mlp_extractor.value_net.2.weight')):
        if a not in sd:
            raise ValueError(f'{path}: no {a} in policy.pth -- not a CustomActorCriticPolicy checkpoint')
        if b in sd and not torch.equal(sd[a], sd[b]):
            raise ValueError(f'{path}: policy and value trunks differ; the fused policy kernel implements the shared trunk only')
    if any(k.startswith('mlp_extractor.policy_net.4') for k in sd):
        raise ValueError(f'{path}: more than two hidden layers')
    return {k: sd[v] for k, v in _KEYS.items()}


def load_policy_zip(path, **kw):
    from .policy import HipPolicy
    t = read_policy_zip(path)
    pol = HipPolicy(obs_dim=t['w1'].shape[1], act_dim=t['wa'].shape[0], hidden=t['w1'].shape[0], **kw)
    pol.load_state(**t)
    return pol


def write_policy_zip(policy, path, data=None):
    """Tensors of a HipPolicy (or any object with w1..log_std) -> a zip with `policy.pth` under SB3 1.0's member and key names (the
    weights are interchangeable; the archive as a whole is not a complete PPO.load input, see the module docstring)."""
    sd = {}
    for k, name in _KEYS.items():
        sd[name] = getattr(policy, k).detach().cpu()
    for layer in ('0', '2'):       # the shared modules appear under both names in the reference's state dict
        for p in ('weight', 'bias'):
            sd[f'mlp_extractor.value_net.{layer}.{p}'] = sd[f'mlp_extractor.policy_net.{layer}.{p}']
    buf = io.BytesIO(); torch.save(sd, buf)
    empty = io.BytesIO(); torch.save({}, empty)
    meta = dict(policy_class='CustomActorCriticPolicy', policy_kwargs=dict(log_std_init=float(getattr(policy, 'log_std').detach()[0])))
    meta.update(data or {})
    with zipfile.ZipFile(path, 'w') as z:
        z.writestr('data', json.dumps(meta, indent=4))
        z.writestr('policy.pth', buf.getvalue())
        z.writestr('pytorch_variables.pth', empty.getvalue())
        z.writestr('_stable_baselines3_version', '1.0')


# ---- the whole archive in SB3 1.0's save_to_zip_file layout -------------------------------------------
_PARAM_ORDER = ('log_std', 'w1', 'b1', 'w2', 'b2', 'wa', 'ba', 'wv', 'bv')      # ActorCriticPolicy.parameters(): log_std, mlp_extractor (shared trunk once), action_net, value_net
_POLICY_CLASS = ('drloco.custom.policies', 'CustomActorCriticPolicy')


def _serialized(obj):
    """One non-JSON entry of `data` as SB3 1.0's data_to_json writes it."""
    import base64
    import cloudpickle
    d = {':type:': str(type(obj)), ':serialized:': base64.b64encode(cloudpickle.dumps(obj)).decode()}
    if hasattr(obj, '__dict__'):
        for k, v in obj.__dict__.items():
            d[k] = str(v)
    return d


def _placeholder_class(module, name):
    """The class `module.name` if importable, else a stand-in registered under that path so that the pickle stream names the real
    class (by reference); returns (class, was_placeholder)."""
    try:
        __import__(module)
        mod = sys.modules[module]
        return getattr(mod, name), not hasattr(mod, '__file__')          # (a module without a file is one of our own placeholders)
    except Exception:
        parts = module.split('.')
        for i in range(1, len(parts) + 1):
            sys.modules.setdefault('.'.join(parts[:i]), types.ModuleType('.'.join(parts[:i])))
        cls = type(name, (object,), {'__module__': module})
        setattr(sys.modules[module], name, cls)
        return cls, True


def _drop_placeholder_modules(roots):
    for m in [m for m in sys.modules if m.split('.')[0] in roots and not hasattr(sys.modules[m], '__file__')]:
        del sys.modules[m]


def _gym_box(low, high):
    """gym.spaces.Box(low, high) -- the real class when gym imports, else an attribute-compatible object pickled under gym's class path."""
    cls, fake = _placeholder_class('gym.spaces.box', 'Box')
    low, high = np.asarray(low, np.float32), np.asarray(high, np.float32)
    if not fake:
        return cls(low=low, high=high, dtype=np.float32), fake
    o = cls.__new__(cls)
    o.__dict__.update(dtype=np.dtype(np.float32), shape=low.shape, low=low, high=high, bounded_below=-np.inf < low, bounded_above=np.inf > high, np_random=None)
    return o, fake


def adam_state_dict(policy, optimizer=None, lr=5e-4):
    """`policy.optimizer.pth`: the state dict of torch.optim.Adam over the policy's nine parameters in SB3's order.  optimizer: a torch
    optimiser whose single param group holds those tensors (in any order; matched by identity) or None = a fresh Adam (no moments yet)."""
    params = [getattr(policy, k) for k in _PARAM_ORDER]
    group = dict(lr=lr, betas=(0.9, 0.999), eps=1e-5, weight_decay=0, amsgrad=False)          # SB3 1.0 PPO: Adam(eps=1e-5)
    state = {}
    if optimizer is not None:
        g0 = optimizer.param_groups[0]
        group.update({k: g0[k] for k in ('lr', 'betas', 'eps', 'weight_decay', 'amsgrad') if k in g0})
        for i, p in enumerate(params):
            for q, st in optimizer.state.items():
                if q is p:
                    state[i] = {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    group['params'] = list(range(len(params)))
    return {'state': state, 'param_groups': [group]}


def write_model_zip(policy, path, observation_space=None, action_space=None, optimizer=None, data=None, hyper=None):
    """The reference's `model.save(path)` (drloco/common/utils.py:175-192; callback.py:290) for a HipPolicy: SB3 1.0's save_to_zip_file
    layout.  observation_space / action_space: (low, high) pairs or objects with .low / .high (HipVecEnv's spaces); default: the straight
    walker's.  hyper: PPO's scalar constructor arguments for `data` (defaults: the reference's, drloco/config/hypers.py + train.py:110-118; `log_std_init`
    is the policy's constructor value, `lr_start` / `lr_final` the end points of the reference's linear decay)."""
    sd = {}
    for k, name in _KEYS.items():
        sd[name] = getattr(policy, k).detach().cpu()
    for layer in ('0', '2'):
        for q in ('weight', 'bias'):
            sd[f'mlp_extractor.value_net.{layer}.{q}'] = sd[f'mlp_extractor.policy_net.{layer}.{q}']
    lohi = lambda sp, d: (np.asarray(sp.low), np.asarray(sp.high)) if hasattr(sp, 'low') else (d if sp is None else sp)
    obs_dim, act_dim = policy.w1.shape[1], policy.wa.shape[0]
    olo, ohi = lohi(observation_space, (np.full(obs_dim, -np.inf), np.full(obs_dim, np.inf)))
    alo, ahi = lohi(action_space, (np.full(act_dim, -300.0), np.full(act_dim, 300.0)))
    fakes = []
    try:
        obs_box, f1 = _gym_box(olo, ohi)
        act_box, f2 = _gym_box(alo, ahi)
        pol_cls, f3 = _placeholder_class(*_POLICY_CLASS)
        fakes = [f1 or f2, f3]
        # the reference's PPO(...) call (drloco/train.py:110-118 with drloco/config/hypers.py:68-116): n_steps = batch_size // n_envs = 16384 // 8,
        # minibatch 2048, clip_range_vf = clip_range, learning rate = LinearDecay(lr_start = 5e-4 -> lr_final = 1e-6) of the remaining progress
        # (drloco/common/schedules.py:16-27).  SB3 stores the schedule as a pickled callable; here its two end points travel as plain values
        # (`lr_start`, `lr_final`) next to `learning_rate` = the value at the current progress, so that a resumed run can rebuild the decay.
        h = dict(learning_rate=5e-4, lr_start=5e-4, lr_final=1e-6, gamma=0.995, gae_lambda=0.95, n_steps=2048, batch_size=2048, n_epochs=4, ent_coef=-0.0075, vf_coef=0.5,
                 max_grad_norm=0.5, clip_range=0.15, clip_range_vf=0.15, n_envs=8, num_timesteps=0, seed=None, verbose=1, sde_sample_freq=-1, use_sde=False, target_kl=None,
                 tensorboard_log=None, log_std_init=-0.75)
        h['log_std_init'] = float(getattr(policy, 'log_std_init', h['log_std_init']))          # the policy's own constructor value
        h.update(hyper or {})
        if (h['n_steps'] * h['n_envs']) % h['batch_size']:
            # as SB3's PPO does: a warning, never an exception -- this runs AFTER training and must not lose the model
            import warnings
            warnings.warn(f"n_steps * n_envs = {h['n_steps'] * h['n_envs']} is not a multiple of the minibatch size {h['batch_size']}: the last minibatch of an epoch is truncated "
                          '(the reference uses 2048 x 8 / 2048)')
        meta = dict(h)
        meta['policy_class'] = _serialized(pol_cls)
        meta['policy_kwargs'] = dict(log_std_init=float(meta.pop('log_std_init')))          # the CONSTRUCTOR value (hypers.init_logstd), not the trained log_std (that is in policy.pth)
        meta['observation_space'] = _serialized(obs_box)
        meta['action_space'] = _serialized(act_box)
        meta.update(data or {})
        bufs = {}
        for name, obj in (('policy.pth', sd), ('policy.optimizer.pth', adam_state_dict(policy, optimizer, lr=h['learning_rate'])), ('pytorch_variables.pth', {})):
            b = io.BytesIO(); torch.save(obj, b); bufs[name] = b.getvalue()
        with zipfile.ZipFile(path, 'w') as z:
            z.writestr('data', json.dumps(meta, indent=4))
            for name, raw in bufs.items():
                z.writestr(name, raw)
            z.writestr('_stable_baselines3_version', '1.0')
    finally:
        if fakes and fakes[0]:
            _drop_placeholder_modules({'gym'})
        if fakes and len(fakes) > 1 and fakes[1]:
            _drop_placeholder_modules({'drloco'})


# ---- the packaged walking policy (drloco_amd/data/walking_policy.npz) ----------------------------------------------------
WALKING_POLICY = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'walking_policy.npz')
# count_steps_same_vel of a walker that has been training for a while (quirk Q2: the counter never resets; 8 M steps / 128 walkers / ~132 control steps per
# reference step): any value >= the number of steps makes the desired-velocity observation the constant step_velocities[0] the policy was trained on
TRAINED_STEP_COUNT = 457


def load_walking_policy(path=None, vec_normalize=None, **policy_kw):
    """The packaged TRAINED policy (examples/train_ppo.py, the reference's 8 M-step budget: the walker reaches the 3000-step episode limit and walks ~22 m per
    episode -- what "trained" means in drloco/common/callback.py:336-369) -> HipPolicy; with `vec_normalize` (a HipVecNormalize) also its moments and, for the walkers of
    its env, the step counter of a training env (quirk Q2: see evaluation.make_eval_env(history='training')).  tools/pack_walking_ckpt.py writes the file from a
    `train_ppo.py --save` checkpoint.  Returns (policy, meta dict)."""
    import json
    from .policy import HipPolicy
    with np.load(path or WALKING_POLICY) as z:
        g = {k: z[k] for k in z.files}
    pol = HipPolicy(obs_dim=g['w1'].shape[1], act_dim=g['wa'].shape[0], hidden=g['w1'].shape[0], **policy_kw)
    pol.load_state(**{k: torch.as_tensor(g[k]) for k in _KEYS})
    meta = json.loads(str(g['meta']))
    if vec_normalize is not None:
        vn = vec_normalize
        vn.obs_rms.load_state(dict(mean=g['obs_mean'], var=g['obs_var'], count=float(g['obs_count'])))
        vn.ret_rms.load_state(dict(mean=g['ret_mean'], var=g['ret_var'], count=float(g['ret_count'])))
        st = vn.venv.get_state()
        st['cursor'][abi.DL_CUR_COUNT] = TRAINED_STEP_COUNT
        vn.venv.set_state(cursor=st['cursor'])
    return pol, meta


def moment_seat(vec_normalize):
    """restore() = put a HipVecNormalize's observation / return moments back to what they are NOW (call right after load_walking_policy).  Why: a FIXED policy under
    VecNormalize statistics that keep training leaves its own input distribution -- the checkpoint's variances hold the whole training history (falls included); fed with
    walking-only data they shrink, the normalised observations grow, and after ~100 rollouts the walkers fall as often as under random torques (measured in bench.py:
    episode length 1610 after 18 rollouts, 179 after 150).  In training the policy moves with its statistics; a benchmark or diagnostic that repeats rollouts with a fixed
    policy calls restore() before each one: every rollout then starts from the checkpoint's moments and advances them by its own samples (no work of the path is skipped)."""
    seat = [(r, r._mean.clone(), r._var.clone(), r._count.clone()) for r in (vec_normalize.obs_rms, vec_normalize.ret_rms)]

    def restore():
        for r, m, v, c in seat:
            r._mean.copy_(m); r._var.copy_(v); r._count.copy_(c)
            for dst, src in zip(r._sync, (m, v, c)):
                dst.copy_(src)
    return restore
