#!/usr/bin/env python3
"""PPO on the MI355X hot path, end to end on one GPU -- a caller of the drop-in surface (the learner stays PyTorch,
as in the reference's train.py; this script is an example, not part of the library).

It follows drloco/train.py:77-139 with the reference's hyperparameters (drloco/config/hypers.py): batch 16 384
samples per update, minibatch 2 048, 4 epochs, gamma 0.995, lambda 0.95, clip range 0.15 (also for the value
function), entropy coefficient -0.0075, learning rate 5e-4 -> 1e-6 (LinearDecay), log_std_init -0.75, 2 x 512 tanh
trunk shared by policy and value heads (drloco/custom/policies.py), Adam eps 1e-5, gradient clipping 0.5.
The default is 128 walkers x 128-step rollouts (the reference: 8 x 2048); rollouts of only a few steps -- e.g.
4096 walkers x 4 steps for the same batch -- learn to stand longer but not to walk, use --batch 262144 --minibatch 16384
with thousands of walkers.  Everything of the rollout runs through the C-ABI:
dl_policy_forward -> dl_step -> dl_vecnormalize_step -> dl_gae; torch autograd only evaluates the PPO loss.

  python examples/train_ppo.py --mio 8          # the reference's budget: 8 M env-steps, ~30 s on one MI355X, walks 23 m per episode (3 of 4 seeds)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ppo.py --envs 1024
                                                # data parallel: walkers shard by global index (--envs is the GLOBAL count), one rank per GPU over RCCL;
                                                # exchanges per update: VecNormalize moment merge (C3), per minibatch [sum a, sum a^2, n] (C1) and ONE
                                                # all-reduce of the flat 1.13 MB gradient bucket (C2) -- drloco_amd/collectives.py; every rank takes the
                                                # optimiser steps one process holding all walkers would take (tests/test_distributed_cpu.py)
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from drloco_amd import collectives
from drloco_amd.policy import HipPolicy
from drloco_amd.rollout import HipRolloutBuffer
from drloco_amd.vec_env import HipVecEnv, HipVecNormalize


def train(mio=2.0, n_envs=128, batch=16384, minibatch=2048, epochs=4, seed=0, log_every=25, quiet=False, norm_reward=True, evaluate=False, save_path=None, moments='per_step', return_objects=False):
    # one process per GPU under torch.distributed.run (backend nccl = RCCL); a single process otherwise
    import torch.distributed as dist
    world, rank, local_rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    dev = torch.device('cuda', local_rank if local_rank < torch.cuda.device_count() else 0)
    torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
        backend = os.environ.get('DL_BENCH_BACKEND', 'nccl')
        dist.init_process_group(backend, rank=rank, world_size=world, **({'device_id': dev} if backend == 'nccl' else {}))
    quiet = quiet or rank != 0
    assert n_envs % world == 0 and batch % n_envs == 0
    if batch % minibatch:          # said at the START of a run (SB3's PPO warns at construction): the last minibatch of every epoch is truncated
        import warnings
        warnings.warn(f'batch {batch} is not a multiple of minibatch {minibatch}: the last minibatch of an epoch has {batch % minibatch} samples')
    n_global, n_envs = n_envs, n_envs // world          # --envs is the global walker count; this rank owns [rank * n_envs, (rank + 1) * n_envs)
    torch.manual_seed(seed)                             # the same initial policy on every rank
    venv = HipVecEnv(num_envs=n_envs, seed=seed, device=dev.index, env_index_base=rank * n_envs)
    vn = HipVecNormalize(venv, norm_reward=norm_reward)
    T = batch // n_global
    buf = HipRolloutBuffer(T, n_envs, venv.obs_dim, venv.nu, dev, gamma=0.995, gae_lambda=0.95)
    pol = HipPolicy(obs_dim=venv.obs_dim, act_dim=venv.nu, hidden=512, log_std_init=-0.75, seed=seed, index_base=rank * n_envs, device=dev)
    names = ('w1', 'b1', 'w2', 'b2', 'wa', 'ba', 'wv', 'bv', 'log_std')
    # SB3's orthogonal initialisation (gain sqrt(2) trunk, 0.01 action head, 1 value head)
    for n_, g in (('w1', math.sqrt(2)), ('w2', math.sqrt(2)), ('wa', 0.01), ('wv', 1.0)):
        torch.nn.init.orthogonal_(getattr(pol, n_), gain=g)
    for n_ in ('b1', 'b2', 'ba', 'bv'):
        getattr(pol, n_).zero_()
    params = [getattr(pol, n_).requires_grad_(True) for n_ in names]
    wd = dict(zip(names, params))
    bucket = collectives.FlatGradAllReducer(params)      # C2: one all-reduce of the flat gradient per optimiser step
    perm_gen = torch.Generator().manual_seed(seed + 1000)    # minibatch permutations: the same on every rank
    opt = torch.optim.Adam(params, lr=5e-4, eps=1e-5)
    clip, ent_coef, vf_coef = 0.15, -0.0075, 0.5
    total = int(mio * 1e6)
    n_updates = max(1, total // batch)
    vn.reset()
    obs = vn.norm_obs_t.clone()
    start = torch.ones(n_envs, dtype=torch.uint8, device=dev)
    last_done = torch.zeros(n_envs, dtype=torch.uint8, device=dev)
    hist = []
    t0 = time.perf_counter()
    for upd in range(n_updates):
        lr = max(5e-4 + (upd / n_updates) * (1e-6 - 5e-4), 1e-6)             # LinearDecay(5e-4 -> 1e-6)
        for gp in opt.param_groups:
            gp['lr'] = lr
        # ---- collect_rollouts
        with torch.no_grad():
            buf.collect_rollouts(vn, pol, obs, start, moments=moments)        # dl_collect_rollouts: T x (policy -> step -> normalise) in one call -- one persistent launch where that form exists
            last_done.copy_(start)
            _, last_values, _ = pol.forward(obs, deterministic=True)
            buf.compute_returns_and_advantage(last_values, last_done)
            if world > 1:
                vn.sync_moments()                            # C3: every rank normalises with the moments of all walkers
        # ---- PPO.train
        b_obs = buf.observations.reshape(-1, venv.obs_dim); b_act = buf.actions.reshape(-1, venv.nu)
        b_adv = buf.advantages.reshape(-1); b_ret = buf.returns.reshape(-1); b_val = buf.values.reshape(-1); b_lp = buf.log_probs.reshape(-1)
        for ep in range(epochs):
            perm = torch.randperm(batch, generator=perm_gen).to(dev)          # indices t * n_global + i into the rollout of ALL walkers
            for i in range(0, batch, minibatch):
                idx = collectives.shard_minibatch(perm[i:i + minibatch], n_global, rank, world)      # this rank's samples of the global minibatch
                adv, n_mb = collectives.minibatch_adv_normalize(b_adv[idx])                          # C1: statistics of the global minibatch
                loss = collectives.ppo_minibatch_loss(wd, b_obs[idx], b_act[idx], adv, b_ret[idx], b_val[idx], b_lp[idx], n_mb, clip, ent_coef, vf_coef)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                bucket.reduce()
                torch.nn.utils.clip_grad_norm_(params, 0.5)
                opt.step()
        if upd % log_every == 0 or upd == n_updates - 1:
            torch.cuda.synchronize()
            ep_len = torch.tensor(venv.get_attr('ep_len_smoothed')).mean().item()
            mean_rew = torch.tensor(venv.get_attr('mean_reward_smoothed')).mean().item()
            dist_ = torch.tensor(venv.get_attr('moved_distance')).mean().item()
            el = time.perf_counter() - t0
            hist.append(dict(update=upd, env_steps=(upd + 1) * batch, ep_len=ep_len, mean_step_reward=mean_rew, moved_distance=dist_, seconds=el, obs_var_max=float(vn.obs_rms.var.max())))
            if not quiet:
                print(f'update {upd:5d}  env-steps {(upd + 1) * batch / 1e6:6.2f} M  ep_len {ep_len:7.1f}  step reward {mean_rew:.3f}  '
                      f'walked {dist_:5.2f} m  max obs var {float(vn.obs_rms.var.max()):9.3g}  lr {lr:.2e}  {el:6.1f} s  ({(upd + 1) * batch / el / 1e3:.0f} k env-steps/s incl. learning)', flush=True)
    if evaluate:
        # TrainingMonitor.eval_walking (drloco/common/callback.py:272-390): 20 deterministic episodes, here as one batch
        from drloco_amd.evaluation import evaluate_walking, make_eval_env
        # Two protocols, both recorded: 'fresh' is the reference's (load_env builds a NEW environment, count_steps_same_vel = 1) and the
        # reference-comparable headline; 'training' carries the training walkers' step counter into the evaluation env -- a fresh env's
        # desired-velocity observation is one the policy has not seen since its first thousand steps (drloco_amd/evaluation.py).  Episodes with an
        # odd evaluation counter k start mirrored against the reference data they read (quirk Q3 of _get_deterministic_init_state) and fall.
        res = evaluate_walking(make_eval_env(vn, history='fresh'), pol)
        res['history'] = 'fresh'
        res_t = evaluate_walking(make_eval_env(vn, history='training'), pol)
        res_t['history'] = 'training'
        hist[-1]['evaluation'] = res                               # the reference's protocol
        hist[-1]['evaluation_training_history'] = res_t            # NOT comparable with the reference's eval numbers
        if not quiet:
            import numpy as np
            for r, label in ((res, "reference protocol: fresh env"), (res_t, "training step counter carried over -- not the reference's protocol")):
                dist_k = np.array(r['moved_distances'])
                print(f"evaluation (20 deterministic episodes, {label}): mean distance {r['mean_walked_distance']:.1f} m (even k {dist_k[0::2].mean():.1f} m, odd k {dist_k[1::2].mean():.1f} m), "
                      f"mean episode length {r['mean_episode_duration'] * 3000:.0f}, stable walks {r['count_stable_walks']}/20, "
                      f"mean step reward (normalised) {r['mean_reward_means']:.2f}")
    if save_path:
        # utils.save_model (drloco/common/utils.py:175-192): models/model_<ckpt>.zip (policy.pth under SB3 1.0's key names) + envs/env_<ckpt> (VecNormalize statistics)
        from drloco_amd import checkpoint
        os.makedirs(os.path.join(save_path, 'models'), exist_ok=True); os.makedirs(os.path.join(save_path, 'envs'), exist_ok=True)
        ckpt = f'{int(total / 1e5)}'
        # SB3 1.0's save_to_zip_file layout: data (spaces, policy class, hyperparameters), policy.pth, policy.optimizer.pth (Adam's moments)
        checkpoint.write_model_zip(pol, os.path.join(save_path, 'models', f'model_{ckpt}.zip'), observation_space=venv.observation_space, action_space=venv.action_space, optimizer=opt,
                                   hyper=dict(n_envs=n_envs, num_timesteps=int(total), n_steps=batch // n_envs, batch_size=minibatch, n_epochs=epochs, learning_rate=float(lr)))
        vn.save(os.path.join(save_path, 'envs', f'env_{ckpt}'), sb3_format=True)
        if not quiet:
            print('saved', os.path.join(save_path, 'models', f'model_{ckpt}.zip'), 'and', os.path.join(save_path, 'envs', f'env_{ckpt}'))
    if return_objects:          # (tools/diag_eval.py)
        return hist, pol, vn
    return hist


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--mio', type=float, default=2.0, help='million env-steps (the reference trains 8)')
    ap.add_argument('--envs', type=int, default=128, help='parallel walkers (the reference: 8)')
    ap.add_argument('--batch', type=int, default=16384, help='samples per update = envs x rollout steps (the reference: 16 384)')
    ap.add_argument('--minibatch', type=int, default=2048)
    ap.add_argument('--seed', type=int, default=1, help='seeds 1, 2, 3 learn to walk within 8 M steps with the current kernels, seed 0 plateaus (DESIGN.md 5.1)')
    ap.add_argument('--save', default=None, help='directory for models/model_<ckpt>.zip (policy.pth with SB3 1.0 key names) and envs/env_<ckpt> (VecNormalize statistics), drloco_amd/checkpoint.py')
    ap.add_argument('--no-norm-reward', action='store_true', help='VecNormalize(norm_reward=False)')
    ap.add_argument('--moments', choices=['per_step', 'per_rollout'], default='per_step', help="VecNormalize moments inside a rollout: SB3's per-step update (default) or the opt-in per-rollout relaxation of the persistent rollout kernel")
    args = ap.parse_args()
    train(args.mio, args.envs, batch=args.batch, minibatch=args.minibatch, seed=args.seed, norm_reward=not args.no_norm_reward, evaluate=True, save_path=args.save, moments=args.moments)
