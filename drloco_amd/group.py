"""Several env handles on one GPU, each on its own HIP stream and with its own policy -> env step -> VecNormalize chain.

Idea: with one wave per SIMD a single-step launch of the 16-lane env kernel lasts as long as its slowest wave, 1.5 x the
mean wave.  `dl_rollout_fixed` hides that by running many control steps per launch, which a policy in
the loop forbids.  Oversubscription hides it as well: two handles of 4096 walkers stepping on two streams reach 25.4 M
env-steps/s with single-step launches (tools/diag_streams.py; the workgroups of one handle's step kernel fill the SIMDs
the other handle's fast waves have left), against 17.8 M for one handle of 4096.

Measured WITH the policy in the loop (bench.py --policy --handles H, one MI355X, round 2): it pays as soon as there is more than
one wave per SIMD -- 8192 walkers: 18.1 M env-steps/s as 2 handles against 16.5 M as one; 32 768: 21.3 M against 20.0 M; 65 536 as 4
handles 21.8 M.  At 4096 walkers it does not (9.7 M as 2 handles against 14.5 M): a half-size launch still has one wave per SIMD it
occupies and lasts as long as its slowest wave, so two half chains just alternate.  Round 1 saw no gain at any
size because `k_policy_forward` needed 74 KB of LDS per workgroup and convoyed behind the resident step kernels; it needs 23 KB
now and fits next to four of their workgroups per CU.

Semantics: the handles are shards exactly like the ranks of a multi-GPU run: walker i of handle h is
global walker `index_base + h * n_per_handle + i` for the RSI and policy-noise streams, so every handle's rollout is
bit for bit what that shard produces on its own (tests/test_gpu_parity.py::test_env_group_handles_are_shards); every
handle normalises with its own VecNormalize moments inside a rollout and `sync_moments()` merges them exactly between
rollouts (the same relaxation as across ranks); the advantage statistics are summed over the handles (and all-reduced
over ranks) before normalising.

Streams: chains only overlap if their streams sit on different hardware queues; two default-priority torch streams
may share one (measured: no overlap), so consecutive handles alternate between the two stream priorities."""
import ctypes as C

import torch

from . import collectives, lib, models
from .rollout import HipRolloutBuffer
from .vec_env import HipVecEnv, HipVecNormalize, _ptr


class HipEnvGroup:
    def __init__(self, n_steps, env_id=models.STRAIGHT_WALKER, num_envs=8192, handles=2, device=None, seed=33, index_base=0,
                 gamma=0.995, gae_lambda=0.95, norm_reward=True, **env_kw):
        assert num_envs % handles == 0
        self.T, self.H, self.n = int(n_steps), int(handles), num_envs // handles
        self.num_envs = num_envs
        self.venvs = [HipVecEnv(env_id, num_envs=self.n, device=device, seed=seed, env_index_base=index_base + h * self.n, **env_kw) for h in range(self.H)]
        dev = self.venvs[0].device
        self.device = dev
        self.vns = [HipVecNormalize(v, norm_reward=norm_reward) for v in self.venvs]
        self.bufs = [HipRolloutBuffer(self.T, self.n, v.obs_dim, v.nu, dev, gamma=gamma, gae_lambda=gae_lambda) for v in self.venvs]
        self.streams = [torch.cuda.Stream(device=dev, priority=-(h % 2)) for h in range(self.H)]
        self.index_bases = [index_base + h * self.n for h in range(self.H)]
        self.last_obs, self.last_done = [], []
        for vn in self.vns:
            vn.reset()
            self.last_obs.append(vn.norm_obs_t.clone())
            self.last_done.append(torch.ones(self.n, dtype=torch.uint8, device=dev))
        torch.cuda.current_stream().synchronize()

    # the handles' chains, interleaved in short runs so that no stream starves while the host enqueues the others
    def collect_rollouts(self, policy, chunk=8):
        """SB3 collect_rollouts for every handle: T x (policy forward -> env step -> VecNormalize) per handle through
        dl_rollout_policy, `chunk` steps at a time round-robin over the handles' streams.  `policy`: a HipPolicy; all handles
        read its weights, their noise streams are keyed by the global walker index.  Returns after ENQUEUEING; `join()` waits."""
        main = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(main)
        counter0 = policy.counter
        pp = policy._params()
        for t0 in range(0, self.T, chunk):
            k = min(chunk, self.T - t0)
            for h in range(self.H):
                b, vn = self.bufs[h], self.vns[h]
                with torch.cuda.stream(self.streams[h]):
                    if t0 == 0:
                        b.reset()
                        b.observations[0].copy_(self.last_obs[h]); b.episode_starts[0].copy_(self.last_done[h])
                    st = vn.state_struct()
                    nxt_obs = b.observations[t0 + k] if t0 + k < self.T else self.last_obs[h]
                    nxt_done = b._starts[t0 + k] if t0 + k < self.T else self.last_done[h]
                    lib.check(b._lib.dl_rollout_policy(vn.venv._h, C.byref(pp), policy.seed, counter0 + t0, self.index_bases[h], C.byref(st), k,
                                                       _ptr(b.observations[t0:]), _ptr(b.actions[t0:]), _ptr(b.values[t0:]), _ptr(b.log_probs[t0:]),
                                                       _ptr(b.rewards[t0:]), _ptr(b._starts[t0:]), _ptr(nxt_obs), _ptr(nxt_done), _ptr(vn.venv.obs), _ptr(vn.venv.rew),
                                                       C.c_void_p(self.streams[h].cuda_stream)))
                    b.pos = t0 + k
        policy.counter = counter0 + self.T

    def join(self):
        main = torch.cuda.current_stream()
        for s in self.streams:
            main.wait_stream(s)

    def compute_returns_and_advantage(self, policy, process_group=None):
        """Per handle: value of the observation that follows the rollout (deterministic forward), GAE; then the advantage
        normalisation over ALL handles (and ranks): sums of the handles added, all-reduced, applied to every handle."""
        for h in range(self.H):
            with torch.cuda.stream(self.streams[h]):
                _, last_values, _ = policy_forward_values(policy, self.last_obs[h], self.index_bases[h])
                self.bufs[h].compute_returns_and_advantage(last_values, self.last_done[h])
                self.bufs[h].advantage_sums()
        self.join()
        sums = torch.stack([b._sums for b in self.bufs]).sum(0)
        collectives.all_reduce_sum_(sums, process_group)
        for b in self.bufs:
            lib.check(b._lib.dl_adv_normalize(_ptr(b.advantages), b.advantages.numel(), _ptr(sums), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def sync_moments(self, process_group=None):
        """Merge the handles' VecNormalize moments exactly (each holds the state agreed at the last merge advanced by its own
        batches): increments [n, sum, sum of squares] summed over the handles (and all-reduced over ranks), applied to all."""
        self.join()
        for name in ('obs_rms', 'ret_rms'):
            merge_moments_local([getattr(vn, name) for vn in self.vns], process_group)

    # what a learner reads: the handles' buffers side by side along the walker axis
    def cat(self, name):
        return torch.cat([getattr(b, name) for b in self.bufs], dim=1)

    def close(self):
        for v in self.venvs:
            v.close()


def policy_forward_values(policy, obs, index_base):
    """Deterministic forward for the bootstrap values (SB3: policy.forward on the last observation); does not advance the
    policy's sampling counter."""
    c = policy.counter
    keep = policy.index_base
    policy.index_base = index_base
    out = policy.forward(obs, deterministic=True)
    policy.counter, policy.index_base = c, keep
    return out


def merge_moments_local(rms_list, process_group=None):
    """Exact merge of several RunningMeanStd objects of one process (plus the other ranks' when distributed): the same
    arithmetic as vec_env.merge_moments_across_ranks, with the increments summed over the local objects first."""
    import torch.distributed as dist
    incs = []
    for r in rms_list:
        mean0, var0, n0 = r._sync
        n1 = r._count
        incs.append(torch.cat([(n1 * r._mean - n0 * mean0).reshape(-1), (n1 * (r._var + r._mean * r._mean) - n0 * (var0 + mean0 * mean0)).reshape(-1), (n1 - n0).reshape(-1)]))
    inc = torch.stack(incs).sum(0)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(inc, group=process_group)
    mean0, var0, n0 = rms_list[0]._sync                  # all objects agreed on this state at the last merge
    d = mean0.numel()
    n = n0 + inc[2 * d]
    s = n0 * mean0 + inc[:d].reshape(mean0.shape)
    q = n0 * (var0 + mean0 * mean0) + inc[d:2 * d].reshape(mean0.shape)
    new_mean = s / n
    new_var = torch.clamp(q / n - new_mean * new_mean, min=0)
    for r in rms_list:
        r._mean.copy_(new_mean); r._var.copy_(new_var); r._count.copy_(n.reshape(r._count.shape))
        r._sync = (r._mean.clone(), r._var.clone(), r._count.clone())
