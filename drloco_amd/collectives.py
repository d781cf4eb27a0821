"""The three exchanges of the data-parallel path, device-agnostic so that the same code runs over RCCL on
the GPUs and over gloo in the world-size-2 CPU tests (tests/test_distributed_cpu.py):

  C1  advantage-normalisation statistics: all-reduce of [sum a, sum a^2, n] (3 doubles) -- once per rollout for the
      rollout-level normalisation of HipRolloutBuffer, once per minibatch for SB3's per-minibatch normalisation
      (PPO.train, constructed at drloco/train.py:110-118);
  C2  gradient step: ONE all-reduce of the flat gradient bucket per optimiser step (282 641 parameters = 1.13 MB for the
      reference's 29-512-512-{8,1} network, 32 steps per update); every rank then clips and steps alike;
  C3  VecNormalize moments: merge_moments_across_ranks (drloco_amd/vec_env.py), once per rollout.

Walkers shard by contiguous global index ranges; a global minibatch is a set of global sample indices of which every rank
takes the samples of its own walkers (`shard_minibatch`), losses are local sums divided by the GLOBAL count, so the summed
gradients are exactly those of one process holding all walkers."""
import math

import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_(t, group=None):
    """In-place sum over ranks (a no-op for a single process)."""
    if world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t


# ---- C1 ---------------------------------------------------------------------------------------------------------
def normalize_advantages(adv, stats_fn, apply_fn, group=None):
    """(A - mean) / (std_unbiased + 1e-8) over ALL ranks.  stats_fn(adv) -> float64[3] = [sum, sum of squares, n] of the
    local part (dl_adv_stats on the device), apply_fn(adv, sums) normalises in place from the global sums (dl_adv_normalize)."""
    sums = stats_fn(adv)
    all_reduce_sum_(sums, group)
    apply_fn(adv, sums)
    return adv


def torch_adv_stats(adv):
    a = adv.double()
    return torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), dtype=torch.float64, device=a.device)])


def torch_adv_apply(adv, sums):
    cnt, mean = sums[2], sums[0] / sums[2]
    var = torch.clamp((sums[1] - cnt * mean * mean) / (cnt - 1), min=0)
    adv.copy_(((adv.double() - mean) / (torch.sqrt(var) + 1e-8)).to(adv.dtype))


def minibatch_adv_normalize(adv_local, group=None):
    """SB3's per-minibatch normalisation with the statistics of the GLOBAL minibatch; returns (normalised local part, n_global
    as a float64 tensor on the data's device -- no host round trip)."""
    s = torch_adv_stats(adv_local)
    all_reduce_sum_(s, group)
    cnt, mean = s[2], s[0] / s[2]
    std = torch.sqrt(torch.clamp((s[1] - cnt * mean * mean) / (cnt - 1), min=0))
    return ((adv_local.double() - mean) / (std + 1e-8)).to(adv_local.dtype), cnt


def shard_minibatch(idx_global, n_global, rank, world):
    """idx_global: flat indices t * n_global + i into the time-major [T, n_global] rollout of ALL walkers (the same permutation
    on every rank).  Returns the flat indices t * n_local + (i - lo) of the samples whose walker belongs to this rank."""
    if n_global % world:
        raise ValueError(f'shard_minibatch: {n_global} walkers do not divide over {world} ranks (contiguous equal index ranges, DESIGN.md 6); '
                         'the samples of the remainder would silently drop out of every minibatch')
    n_local = n_global // world
    lo = rank * n_local
    t, i = idx_global // n_global, idx_global % n_global
    mine = (i >= lo) & (i < lo + n_local)
    return t[mine] * n_local + (i[mine] - lo)


# ---- C2 ---------------------------------------------------------------------------------------------------------
class FlatGradAllReducer:
    """One contiguous bucket for the gradients of `params`: reduce() packs them, sums over ranks with ONE all-reduce and
    unpacks -- 1.13 MB per optimiser step for the reference's network, latency-bound over xGMI, so one call beats per-tensor
    calls.  Losses are expected to be local sums divided by the global sample count (ppo_minibatch_loss), hence SUM, not mean."""

    def __init__(self, params, group=None):
        self.params, self.group = list(params), group
        p0 = self.params[0]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=p0.dtype, device=p0.device)

    def reduce(self):
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[o:o + n].zero_()
            else:
                self.flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        all_reduce_sum_(self.flat, self.group)
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[o:o + n].reshape(p.shape).clone()
            else:
                p.grad.copy_(self.flat[o:o + n].reshape(p.shape))
            o += n


def ppo_minibatch_loss(w, obs, act, adv, ret, old_val, old_logp, n_global, clip=0.15, ent_coef=-0.0075, vf_coef=0.5):
    """PPO's clipped loss (SB3 1.0 PPO.train with clip_range_vf = clip_range, drloco/train.py:117; network of
    drloco/custom/policies.py:13-51) on the LOCAL part of a minibatch whose global size is n_global: every mean of the
    single-process loss becomes a local sum / n_global, so that gradients summed over ranks are the single-process ones.
    w: dict of tensors w1 b1 w2 b2 wa ba wv bv log_std; adv already normalised (minibatch_adv_normalize)."""
    lin = torch.nn.functional.linear
    if torch.is_tensor(n_global):
        n_global = n_global.to(obs.dtype)
    h = torch.tanh(lin(obs, w['w1'], w['b1']))
    h = torch.tanh(lin(h, w['w2'], w['b2']))
    mean = lin(h, w['wa'], w['ba'])
    value = lin(h, w['wv'], w['bv'])[:, 0]
    std = torch.exp(w['log_std'])
    logp = (-0.5 * ((act - mean) / std) ** 2 - w['log_std'] - 0.5 * math.log(2 * math.pi)).sum(1)
    entropy = (0.5 + 0.5 * math.log(2 * math.pi) + w['log_std']).sum()
    ratio = torch.exp(logp - old_logp)
    share = obs.shape[0] / n_global                     # the entropy term does not depend on the samples: each rank carries its share
    pg_loss = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip, 1 + clip)).sum() / n_global
    v_pred = old_val + torch.clamp(value - old_val, -clip, clip)
    v_loss = ((ret - v_pred) ** 2).sum() / n_global
    return pg_loss + ent_coef * (-entropy) * share + vf_coef * v_loss
