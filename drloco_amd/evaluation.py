"""Batched deterministic evaluation on the device: the counterpart of TrainingMonitor.eval_walking
(drloco/common/callback.py:272-390).

The reference saves model + VecNormalize, reloads both around a 1-env DummyVecEnv, calls
`activate_evaluation()` and plays EVAL_N_TIMES = 20 episodes one after the other with
`predict(obs, deterministic=True)`; episode i starts from `_get_deterministic_init_state` with k = i
(straight_walk_trajecs.py:237-265).  Here the 20 episodes are 20 walkers of one handle: walker i gets the
evaluation counter k = i, and the whole evaluation is ONE dl_collect_rollouts call in its deterministic mode
(DL_ROLLOUT_DETERMINISTIC: the policy returns the mean action; moments frozen) -- one persistent launch where that form
exists -- after which the device's first-episode Monitor words (first_ep_len / first_ep_moved / first_ep_ret) hold what
eval_walking measures per episode.  The step-by-step host loop of earlier rounds stays as `evaluate_walking_host_loop`: the
restatement of the reference's loop the one-call form is tested against.

Differences from the reference loop, on purpose:
  * the evaluation copy of VecNormalize does not keep training its moments (the reference's reloaded copy does,
    because SB3's `VecNormalize.load` leaves `training=True`; serial moment updates cannot be reproduced by a
    batch and the effect is a drift of the 20 episodes' normalisation, not a property anyone relies on);
  * every episode starts from a fresh `count_steps_same_vel` (the reference's counter only ever increments and
    survives resets, straight_walk_trajecs.py:124,336, so in its serial loop the desired-velocity observation of
    episode k depends on how many step rollovers episodes 0..k-1 contained -- history a batch does not have);
  * the per-step reward is the raw reward (`get_original_reward`), which equals the reference's
    `reward * sqrt(ret_rms.var + 1e-8)` un-normalisation whenever the normalised reward was not clipped.
"""
import numpy as np
import torch

from . import abi, lib
from .vec_env import HipVecEnv, HipVecNormalize, _ptr, _stream

EVAL_N_TIMES = 20            # drloco/config/config.py:23
MIN_STABLE_DISTANCE = 15     # drloco/config/config.py:25
CTRL_FREQ = 200              # drloco/config/config.py:20 (StraightMimicWalker)
EP_DUR_MAX = 3000            # drloco/config/hypers.py:58
ALIVE_BONUS, REW_SCALE = 0.2, 1.0   # drloco/config/hypers.py:51-55


def make_eval_env(train_env, n_episodes=EVAL_N_TIMES, history='fresh', **kw):
    """utils.load_env (drloco/common/utils.py:234-240) without the trip through the file system: a fresh handle of
    `n_episodes` walkers in evaluation mode that normalises with a copy of `train_env`'s moments (norm_rew=False as
    in load_env).  Walker i starts its first episode from deterministic init state k = i.
    history: 'fresh' (the reference: load_env builds a NEW environment, whose count_steps_same_vel starts at 1) or 'training' (the counter
    of the training walkers is carried over).  It matters more than it looks: the counter only ever grows and survives resets (quirk Q2,
    straight_walk_trajecs.py:124,336), so after the first few thousand steps of training the desired-velocity observation
    step_velocities[max(0, i_step - count + 1)] is the CONSTANT step_velocities[0]; VecNormalize's variance of that column collapses, and a
    fresh environment -- whose observation walks through the per-step velocities again -- presents the policy with inputs clipped at +-10.
    Measured (tools/diag_eval.py, 6 M steps, seed 1): training episodes 2600 steps / 21 m; 'fresh' evaluation 400-500 steps / 3-4 m;
    'training' history: the training env's behaviour.  The reference's own evaluation has the same mismatch."""
    if history not in ('fresh', 'training'):
        raise ValueError("history must be 'fresh' or 'training'")
    src = train_env.venv
    venv = HipVecEnv(src.env_id, num_envs=n_episodes, device=src.device.index, seed=src.cfg.seed, precision=src.precision,
                     model=src.model, refs=src.refs, **kw)
    vn = HipVecNormalize(venv, training=False, norm_obs=train_env.norm_obs, norm_reward=False, clip_obs=train_env.clip_obs,
                         clip_reward=train_env.clip_reward, gamma=train_env.gamma, epsilon=train_env.epsilon)
    vn.obs_rms.load_state(train_env.obs_rms.state())
    vn.ret_rms.load_state(train_env.ret_rms.state())
    venv.activate_evaluation()
    st = venv.get_state()
    st['cursor'][abi.DL_CUR_EVAL_K] = np.arange(n_episodes) % 20
    if history == 'training':
        st['cursor'][abi.DL_CUR_COUNT] = int(np.median(src.get_state()['cursor'][abi.DL_CUR_COUNT]))
    venv.set_state(cursor=st['cursor'])
    return vn


def _summary(moved, ep_durs, mean_rewards, n, n_saved_models):
    """callback.py:322-345,367-378 from the per-episode numbers"""
    vels = moved / (ep_durs / CTRL_FREQ)
    res = dict(moved_distances=moved.tolist(), ep_durs=ep_durs.tolist(), mean_rewards=mean_rewards.tolist())
    res['mean_walked_distance'], res['min_walked_distance'] = float(np.mean(moved)), float(np.min(moved))
    res['mean_episode_duration'], res['min_episode_duration'] = float(np.mean(ep_durs) / EP_DUR_MAX), int(np.min(ep_durs))
    res['mean_walking_speed'], res['min_walking_speed'] = float(np.mean(vels)), float(np.min(vels))
    res['mean_reward_means'] = float((np.mean(mean_rewards) - ALIVE_BONUS) / REW_SCALE)
    below = np.where(moved < MIN_STABLE_DISTANCE)[0]
    no_fall = np.where((ep_durs == EP_DUR_MAX) & (moved >= 0.5 * MIN_STABLE_DISTANCE))[0]
    res['failed_eval_runs_indices'] = below.tolist()
    res['count_stable_walks'] = int(max(n - len(below), len(no_fall)))
    walks_humanlike = res['mean_reward_means'] >= 0.5 * (1 + n_saved_models / 10)
    res['is_stable_humanlike_walking'] = bool(res['count_stable_walks'] == n and walks_humanlike)
    return res


def evaluate_walking(eval_env, policy, n_saved_models=0, chunk=None, persistent=None):
    """Play the first episode of every walker of `eval_env` (from make_eval_env) with the deterministic policy and return the statistics
    eval_walking computes (same names as the TrainingMonitor attributes, callback.py:322-345,367-378).
    ONE device call: dl_collect_rollouts over `ep_dur_max` control steps with DL_ROLLOUT_DETERMINISTIC (the persistent one-launch form
    where it exists: float32, 16 lanes per walker, hidden = 512 / 256 / 128; three launches per control step enqueued without a host round trip
    otherwise), then three dl_stats_snapshot reads.  chunk: control steps per call (default: the whole episode budget in one call); with a
    smaller chunk the host looks between calls whether every walker has finished and stops early.  `policy`: a HipPolicy; any other object with
    `.forward(obs, deterministic=True)` is evaluated by the step-by-step host loop below.
    A device fault of the persistent form (a hand-over or grid-exchange time-out: another process holding CUs) does not abort a training run that
    evaluates inside its loop: with persistent=None the fault is cleared, the walkers are reset and the evaluation is redone in the launch form, with a
    warning; persistent=True raises DrlocoFault."""
    import ctypes as C
    if not hasattr(policy, '_params'):
        return evaluate_walking_host_loop(eval_env, policy, n_saved_models)
    venv = eval_env.venv
    n, dev = venv.num_envs, venv.device
    eval_env.training = False
    cursor0 = venv.get_state()['cursor'] if persistent is None else None          # (the evaluation counters k of the walkers: a redo after a fault starts from the same init states)
    eval_env.reset()                                                   # (a reset of all walkers also opens a new first-episode record)
    horizon = int(venv.cfg.ep_dur_max)
    T = horizon if chunk is None else max(1, min(int(chunk), horizon))
    f = lambda *s: torch.empty(*s, device=dev)
    obs_buf, act_buf = f(T, n, venv.obs_dim), f(T, n, venv.nu)
    val_buf, lp_buf, rew_buf = f(T, n), f(T, n), f(T, n)
    starts = torch.zeros(T, n, dtype=torch.uint8, device=dev)
    last_obs, last_done = eval_env.norm_obs_t.clone(), torch.ones(n, dtype=torch.uint8, device=dev)
    p, st = policy._params(), eval_env.state_struct()
    ok = bool(venv._lib.dl_rollout_persistent_ok(venv._h, C.byref(p)))
    use_persistent = ok if persistent is None else bool(persistent)
    mode = abi.DL_ROLLOUT_DETERMINISTIC | (abi.DL_ROLLOUT_PERSISTENT if use_persistent else 0)
    first_len = torch.zeros(n, dtype=torch.float64, device=dev)
    steps = 0
    while steps < horizon:
        obs_buf[0].copy_(last_obs); starts[0].copy_(last_done)
        lib.check(venv._lib.dl_collect_rollouts(venv._h, C.byref(p), policy.seed, policy.counter, policy.index_base, C.byref(st), T, _ptr(obs_buf), _ptr(act_buf),
                                                _ptr(val_buf), _ptr(lp_buf), _ptr(rew_buf), _ptr(starts), _ptr(last_obs), _ptr(last_done), _ptr(venv.obs), _ptr(venv.rew),
                                                mode, _stream()))
        policy.counter += T
        steps += T
        if use_persistent:
            torch.cuda.current_stream().synchronize()
            rc = venv._lib.dl_fault_check(venv._h, None)                   # a persistent launch is complete only with a clear fault word
            if rc == abi.DL_E_FAULT and persistent is None:
                import warnings
                warnings.warn('drloco_amd: the persistent evaluation launch reported a device fault (' + venv._lib.dl_last_error().decode() + '); fault cleared, walkers reset, '
                              'evaluation redone with the launch form')
                lib.check(venv._lib.dl_fault_clear(venv._h))
                venv.set_state(cursor=cursor0)
                return evaluate_walking(eval_env, policy, n_saved_models, chunk, persistent=False)
            lib.check(rc)
        lib.check(venv._lib.dl_stats_snapshot(venv._h, b'first_ep_len', _ptr(first_len), _stream()))
        if steps < horizon and bool((first_len > 0).all()):
            break
    moved_t, ret_t = torch.zeros_like(first_len), torch.zeros_like(first_len)
    lib.check(venv._lib.dl_stats_snapshot(venv._h, b'first_ep_moved', _ptr(moved_t), _stream()))
    lib.check(venv._lib.dl_stats_snapshot(venv._h, b'first_ep_ret', _ptr(ret_t), _stream()))
    ep_durs = first_len.cpu().numpy().astype(np.int64)
    if (ep_durs <= 0).any():
        raise lib.DrlocoError('evaluate_walking: a walker has not finished an episode within ep_dur_max control steps')
    moved = moved_t.cpu().numpy()
    with np.errstate(invalid='ignore', divide='ignore'):
        mean_rewards = ret_t.cpu().numpy() / (ep_durs - 1)                 # np.mean(rewards) over the ep_dur - 1 non-terminal steps
    res = _summary(moved, ep_durs, mean_rewards, n, n_saved_models)
    res['device_calls'], res['form'] = steps // T, 'persistent' if use_persistent else 'launches'
    return res


def evaluate_walking_host_loop(eval_env, policy, n_saved_models=0, check_every=250):
    """The same evaluation as a step-by-step host loop (policy forward -> dl_step -> dl_vecnormalize_step, one dl_get_state per step): the
    literal restatement of the reference's loop, kept as the reference `evaluate_walking` is tested against and for policies that are not a
    HipPolicy (`policy.forward(obs, deterministic=True)` -> (actions, values, log_probs) on the device)."""
    venv = eval_env.venv
    n, dev = venv.num_envs, venv.device
    eval_env.training = False
    eval_env.reset()
    obs = eval_env.norm_obs_t
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    ep_dur = torch.zeros(n, dtype=torch.int64, device=dev)
    rew_sum = torch.zeros(n, dtype=torch.float64, device=dev)
    walked = torch.zeros(n, dtype=torch.float64, device=dev)
    walked_now = torch.zeros(n, dtype=torch.float64, device=dev)
    for t in range(EP_DUR_MAX):
        actions, _, _ = policy.forward(obs, deterministic=True)
        obs, _, done, _ = eval_env.step_tensors(actions)
        lib.check(venv._lib.dl_get_state(venv._h, None, None, None, None, _ptr(walked_now), _stream()))
        d = done.bool()
        ep_dur += alive                                     # `ep_dur += 1` happens before the step, the done step included
        cont = alive & ~d                                   # `else` branch of `if done` (callback.py:311-317)
        walked = torch.where(cont, walked_now, walked)      # walked distance / rewards exclude the terminal step
        rew_sum += torch.where(cont, venv.rew.double(), torch.zeros_like(rew_sum))
        alive = cont
        if (t + 1) % check_every == 0 and not bool(alive.any()):
            break
    ep_durs = ep_dur.cpu().numpy()
    moved = walked.cpu().numpy()
    with np.errstate(invalid='ignore', divide='ignore'):
        mean_rewards = rew_sum.cpu().numpy() / (ep_durs - 1)          # np.mean(rewards) over the ep_dur - 1 non-terminal steps
    return _summary(moved, ep_durs, mean_rewards, n, n_saved_models)
