"""Batched deterministic evaluation on the device: the counterpart of TrainingMonitor.eval_walking
(drloco/common/callback.py:272-390).

The reference saves model + VecNormalize, reloads both around a 1-env DummyVecEnv, calls
`activate_evaluation()` and plays EVAL_N_TIMES = 20 episodes one after the other with
`predict(obs, deterministic=True)`; episode i starts from `_get_deterministic_init_state` with k = i
(straight_walk_trajecs.py:237-265).  Here the 20 episodes are 20 walkers of one handle: walker i gets the
evaluation counter k = i, every control step is policy forward -> dl_step -> dl_vecnormalize_step (moments
frozen), nothing returns to the host until all walkers have finished their first episode.

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


def evaluate_walking(eval_env, policy, n_saved_models=0, check_every=250):
    """Play the first episode of every walker of `eval_env` (from make_eval_env) with the deterministic policy and
    return the statistics eval_walking computes (same names as the TrainingMonitor attributes,
    callback.py:322-345,367-378).  `policy.forward(obs, deterministic=True)` -> (actions, values, log_probs) on the
    device (HipPolicy, or any callable object with that method)."""
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
