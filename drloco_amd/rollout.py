"""Device rollout buffer with the SB3 1.0 return/advantage scan
(RolloutBuffer.add / compute_returns_and_advantage, constructed via drloco/train.py:110-118)
and PPO's advantage normalisation, whose statistics are the only rollout-side collective when
walkers are sharded over GPUs (RCCL all-reduce of three doubles)."""
import ctypes as C

import torch

from . import abi, collectives, lib
from .vec_env import _ptr, _stream


class HipRolloutBuffer:
    def __init__(self, n_steps, n_envs, obs_dim, act_dim, device, gamma=0.995, gae_lambda=0.95):
        self._lib = lib.load()
        self.T, self.N, self.gamma, self.gae_lambda = n_steps, n_envs, gamma, gae_lambda
        f = lambda *s: torch.zeros(*s, device=device)
        self.observations, self.actions = f(n_steps, n_envs, obs_dim), f(n_steps, n_envs, act_dim)
        self.rewards, self.values, self.log_probs = f(n_steps, n_envs), f(n_steps, n_envs), f(n_steps, n_envs)
        # one row more than SB3's array: row T takes the done flags of the last step (SB3's _last_episode_starts), so that a
        # run of steps can write its flags into consecutive rows t+1..
        self._starts = torch.zeros(n_steps + 1, n_envs, dtype=torch.uint8, device=device)
        self.episode_starts = self._starts[:n_steps]
        self.next_starts = self._starts[n_steps]
        self.advantages, self.returns = f(n_steps, n_envs), f(n_steps, n_envs)
        self._sums = torch.zeros(3, dtype=torch.float64, device=device)
        self._adv_work = torch.zeros(abi.DL_ADV_WORKSPACE_BYTES // 8, dtype=torch.float64, device=device)     # caller-owned scratch of dl_adv_stats
        self.pos = 0

    def reset(self):
        self.pos = 0

    def add(self, obs, action, reward, episode_start, value, log_prob):
        t = self.pos
        self.observations[t].copy_(obs); self.actions[t].copy_(action); self.rewards[t].copy_(reward)
        self.episode_starts[t].copy_(episode_start); self.values[t].copy_(value); self.log_probs[t].copy_(log_prob)
        self.pos += 1

    def collect_rollouts(self, vn, policy, last_obs, last_done, persistent=None, moments='per_step', workgroup_tiles=False, deterministic=False):
        """SB3 1.0 OnPolicyAlgorithm.collect_rollouts (the loop between two PPO updates) as ONE C-ABI call, dl_collect_rollouts:
        T x (policy forward -> env step -> VecNormalize), every result written straight into this buffer.  vn: HipVecNormalize;
        policy: HipPolicy; last_obs float32 [N, obs] / last_done uint8 [N]: the normalised observation and episode-start flags that
        open this rollout -- overwritten with the ones that open the next (SB3's _last_obs / _last_episode_starts).
        persistent: True = ONE launch for the whole rollout (a persistent workgroup per sixteen walkers, one grid-wide exchange per control
        step; float32, hidden = 512 / 256 / 128, <= 128 walkers per CU (19-dof walker: <= 16 per CU); needs the GPU to itself -- the call waits for the launch and raises DrlocoFault if
        the exchange timed out), False = three launches per control step, None = persistent where it exists, falling back to the launch form
        (walkers reset, moments restored, a warning) if the exchange times out.
        moments: 'per_step' (SB3's semantics, default) or 'per_rollout' (opt-in relaxation, persistent form only: the whole rollout is
        normalised with the moments at its start, which are advanced once, by all T x N samples, at its end -- include/drloco_hip.h).
        workgroup_tiles (with 'per_rollout'): run the relaxation on the exact form's kernel (sixteen-row policy tiles, the workgroup's pairs meet
        every step) instead of the pair-by-pair kernel: the selectable fallback, about 4 % slower (DL_ROLLOUT_WORKGROUP_TILES).
        deterministic: the policy's mean action instead of a sample (DL_ROLLOUT_DETERMINISTIC, every form; not for the cross-rank per-step host loop).

        What the automatic mode (persistent=None) does when the exchange times out, so that nobody is surprised by it: (1) the fault is cleared
        and VecNormalize's moments are restored to their pre-launch snapshot; (2) ALL walkers are reset (vn.reset()): every episode in flight
        is discarded -- the Monitor statistics lose them, and the redone rollout starts from fresh reference-state initialisations instead of
        the states the walkers had reached; (3) the handle's push-schedule clock has advanced by 2 T (the failed launch and the redone one);
        (4) this buffer keeps the launch form from then on (`_persistent_off`): a transient co-tenant downgrades it for good, with only the
        warning as a record -- call `enable_persistent()` to try the persistent form again.  Only DL_FAULT_GRID_TIMEOUT with per-step moments
        takes this path; per-rollout launches and pair / split hand-over faults raise DrlocoFault in every mode."""
        if moments not in ('per_step', 'per_rollout'):
            raise ValueError("moments must be 'per_step' or 'per_rollout'")
        if getattr(vn, 'sync', 'per_rollout') == 'per_step':
            # exact per-step moments ACROSS RANKS (HipVecNormalize(sync='per_step')): every control step's update needs the other ranks' sums, so
            # the loop runs on the host -- policy forward, env step, local sums, all-reduce, merge, normalise: six launches + one collective per
            # control step.  With one rank and blocked_reduce it is the launch form of dl_collect_rollouts bit for bit (tests/test_gpu_persistent.py).
            if deterministic:
                raise lib.DrlocoError("deterministic rollouts go through dl_collect_rollouts: not with HipVecNormalize(sync='per_step')")
            if persistent or moments == 'per_rollout':
                raise lib.DrlocoError("HipVecNormalize(sync='per_step') exchanges moments between ranks every control step: the persistent / per-rollout forms do not apply")
            if vn._ov is not None:
                raise lib.DrlocoError("HipVecNormalize(sync='per_step') and enable_overlap() do not combine (a policy in the loop needs each step's normalised observation)")
            self.reset()
            self.observations[0].copy_(last_obs)
            self.episode_starts[0].copy_(last_done)
            for t in range(self.T):
                policy.forward(self.observations[t], actions_out=self.actions[t], values_out=self.values[t], log_probs_out=self.log_probs[t])
                last = t + 1 == self.T
                vn.step_tensors(self.actions[t], obs_out=last_obs if last else self.observations[t + 1], rew_out=self.rewards[t],
                                done_out=last_done if last else self.episode_starts[t + 1])
            self.pos = self.T
            self.last_form = 'host loop'
            return
        self.reset()
        self.observations[0].copy_(last_obs)
        self.episode_starts[0].copy_(last_done)
        p, st = policy._params(), vn.state_struct()
        ok = bool(self._lib.dl_rollout_persistent_ok(vn.venv._h, C.byref(p)))
        auto = persistent is None
        if auto:
            persistent = ok and not getattr(self, '_persistent_off', False)
        if moments == 'per_rollout' and not persistent:
            raise lib.DrlocoError("moments='per_rollout' exists in the persistent form of collect_rollouts only")

        det = abi.DL_ROLLOUT_DETERMINISTIC if deterministic else 0

        def launch(mode):
            mode |= det
            lib.check(self._lib.dl_collect_rollouts(vn.venv._h, C.byref(p), policy.seed, policy.counter, policy.index_base, C.byref(st), self.T,
                                                    _ptr(self.observations), _ptr(self.actions), _ptr(self.values), _ptr(self.log_probs), _ptr(self.rewards),
                                                    _ptr(self.episode_starts), _ptr(last_obs), _ptr(last_done), _ptr(vn.venv.obs), _ptr(vn.venv.rew), mode, _stream()))
        if persistent:
            # The persistent kernel's grid-wide exchange needs every workgroup co-resident: the GPU has to be this launch's alone (include/drloco_hip.h).
            # Another process or stream holding CUs makes the exchange time out; the kernel then sets the handle's fault word and stops writing -- the call
            # itself has returned DL_OK long before.  So the rollout is complete only once the launch has finished with a clear fault word: wait and look
            # (a rollout lasts tens of milliseconds; the learner needs it finished anyway) instead of handing a half-written buffer to the PPO update.
            snap = [t.clone() for t in (vn.obs_rms._mean, vn.obs_rms._var, vn.obs_rms._count, vn.ret_rms._mean, vn.ret_rms._var, vn.ret_rms._count)] if auto else None
            launch(abi.DL_ROLLOUT_PERSISTENT | (abi.DL_ROLLOUT_MOMENTS_PER_ROLLOUT if moments == 'per_rollout' else 0) |
                   (abi.DL_ROLLOUT_WORKGROUP_TILES if (moments == 'per_rollout' and workgroup_tiles) else 0))
            torch.cuda.current_stream().synchronize()
            code = C.c_int32(0)
            rc = self._lib.dl_fault_check(vn.venv._h, C.byref(code))
            if rc == abi.DL_E_FAULT and auto and (code.value & abi.DL_FAULT_GRID_TIMEOUT) and moments == 'per_step':
                # auto mode: the launch form needs no co-residency.  The half-advanced walkers are re-initialised, the moments restored to their state
                # before the launch, and the rollout is redone in this process with three launches per control step; later rollouts stay on that form.
                import warnings
                warnings.warn('drloco_amd: the persistent rollout kernel timed out in its grid-wide exchange (is another process or stream using this GPU?); '
                              'walkers reset, rollout redone with the launch form, which this buffer keeps using from now on')
                lib.check(self._lib.dl_fault_clear(vn.venv._h))
                for t, s0 in zip((vn.obs_rms._mean, vn.obs_rms._var, vn.obs_rms._count, vn.ret_rms._mean, vn.ret_rms._var, vn.ret_rms._count), snap):
                    t.copy_(s0)
                vn.reset()
                last_obs.copy_(vn.norm_obs_t); last_done.fill_(1)
                self.observations[0].copy_(last_obs); self.episode_starts[0].copy_(last_done)
                self._persistent_off = True
                persistent = False
                launch(0)
            else:
                lib.check(rc)
        else:
            launch(0)
        policy.counter += self.T
        self.pos = self.T
        self.last_form = 'persistent' if persistent else 'launches'

    def enable_persistent(self):
        """Let the automatic mode of collect_rollouts try the persistent form again after a grid-exchange timeout switched it off."""
        self._persistent_off = False

    def compute_returns_and_advantage(self, last_values, dones):
        lv = last_values.to(torch.float32).contiguous()
        ld = dones.to(torch.uint8).contiguous()
        lib.check(self._lib.dl_gae(_ptr(self.rewards), _ptr(self.values), _ptr(self.episode_starts), _ptr(lv), _ptr(ld),
                                   C.c_float(self.gamma), C.c_float(self.gae_lambda), self.T, self.N,
                                   _ptr(self.advantages), _ptr(self.returns), _stream()))
        return self.advantages, self.returns

    def get(self, batch_size=None, generator=None):
        """SB3 1.0 RolloutBuffer.get: minibatches of the flattened rollout in a random order, as RolloutBufferSamples-like
        named tuples (observations, actions, old_values, old_log_prob, advantages, returns) of device tensors.  The rollout is
        flattened env-major as SB3's swap_and_flatten does ([T, N, ...] -> [N * T, ...]); batch_size None = one batch."""
        import collections
        Samples = collections.namedtuple('RolloutBufferSamples', 'observations actions old_values old_log_prob advantages returns')
        flat = lambda x: x.transpose(0, 1).reshape(self.T * self.N, *x.shape[2:])
        cols = [flat(x) for x in (self.observations, self.actions, self.values, self.log_probs, self.advantages, self.returns)]
        total = self.T * self.N
        perm = torch.randperm(total, device=self.observations.device, generator=generator)
        bs = total if batch_size is None else int(batch_size)
        for i in range(0, total, bs):
            idx = perm[i:i + bs]
            yield Samples(*(c[idx] for c in cols))

    def advantage_sums(self, adv=None):
        a = self.advantages if adv is None else adv
        lib.check(self._lib.dl_adv_stats(_ptr(a), a.numel(), _ptr(self._sums), _ptr(self._adv_work), _stream()))
        return self._sums

    def normalize_advantages(self, adv=None, process_group=None):
        """(A - mean)/(std + 1e-8) over all ranks: one all-reduce of [sum, sum^2, n]."""
        a = self.advantages if adv is None else adv
        apply = lambda x, sums: lib.check(self._lib.dl_adv_normalize(_ptr(x), x.numel(), _ptr(sums), _stream()))
        return collectives.normalize_advantages(a, self.advantage_sums, apply, process_group)
