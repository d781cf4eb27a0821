"""HipPolicy: the forward pass of the reference's CustomActorCriticPolicy (drloco/custom/policies.py:13-51, built at
drloco/train.py:105-118) as one fused MFMA kernel (`dl_policy_forward`), for GPU-resident rollouts.

The parameters stay ordinary torch tensors (the PPO update and its optimiser remain PyTorch): `HipPolicy` only reads
them.  `forward(obs)` mirrors SB3 1.0 `ActorCriticPolicy.forward`: (actions, values, log_probs)."""
import ctypes as C
import math

import torch

from . import abi, lib
from .vec_env import _ptr, _stream


class HipPolicy:
    def __init__(self, obs_dim=29, act_dim=8, hidden=512, log_std_init=-0.75, device='cuda', seed=0, index_base=0):
        """Fresh parameters with torch's nn.Linear default initialisation (the reference's CustomHiddenLayers builds
        plain nn.Linear layers; SB3 then applies its orthogonal init -- load trained weights with `load_state`)."""
        self._lib = lib.load()
        g = torch.Generator(device='cpu'); g.manual_seed(seed)

        def linear(o, i):
            bound = 1.0 / math.sqrt(i)
            w = (torch.rand(o, i, generator=g) * 2 - 1) * bound
            b = (torch.rand(o, generator=g) * 2 - 1) * bound
            return w.to(device).contiguous(), b.to(device).contiguous()
        self.obs_dim, self.act_dim, self.hidden = obs_dim, act_dim, hidden
        self.w1, self.b1 = linear(hidden, obs_dim)
        self.w2, self.b2 = linear(hidden, hidden)
        self.wa, self.ba = linear(act_dim, hidden)
        self.wv, self.bv = linear(1, hidden)
        self.log_std_init = float(log_std_init)          # the constructor value (SB3's policy_kwargs; checkpoint.write_model_zip stores it)
        self.log_std = torch.full((act_dim,), float(log_std_init), device=device)
        self.seed, self.counter, self.index_base = int(seed), 0, int(index_base)
        self._packed, self._packed_key, self._packed_stream = None, None, None

    def _packed_weights(self):
        """k-chunk-major copy of the weights for the stand-alone forward pass (hidden = 512 / 256 / 128), refreshed whenever a weight tensor was replaced
        (load_state) or written to (torch bumps `tensor._version` on every in-place update, e.g. an optimiser step) or the caller moved to
        another stream (the copy is written and read in stream order: the key carries the stream, a repack waits for the last read)."""
        if self.hidden not in (512, 256, 128):
            return None
        ws = (self.w1, self.w2, self.wa, self.wv)
        stream = torch.cuda.current_stream()
        try:
            key = tuple((t.data_ptr(), t._version) for t in ws) + (stream.cuda_stream,)
        except RuntimeError:          # inference-mode tensors have no version counter: repack every time
            key = None
        if key is None or key != self._packed_key:
            h = self.hidden
            if self._packed is None or self._packed.numel() != h * h + 48 * h + 16 * h:
                self._packed = torch.empty(h * h + 48 * h + 16 * h, device=self.w1.device)
            if self._packed_stream is not None and self._packed_stream != stream:
                stream.wait_stream(self._packed_stream)          # a forward pass on the other stream may still be reading the old copy
            p = self._params()
            lib.check(self._lib.dl_policy_pack(C.byref(p), _ptr(self._packed), _stream()))
            self._packed_key, self._packed_stream = key, stream
        return self._packed

    def invalidate_packed(self):
        """Call after changing the weights in a way torch's version counter does not see (writes through `.data` or raw pointers)."""
        self._packed_key = None

    def load_state(self, w1, b1, w2, b2, wa, ba, wv, bv, log_std):
        """Take the tensors of a trained torch policy: mlp_extractor.policy_net[0], [2] (shared with value_net),
        action_net, value_net, log_std."""
        f = lambda t: t.detach().to(device=self.w1.device, dtype=torch.float32).contiguous()
        self.w1, self.b1, self.w2, self.b2, self.wa, self.ba, self.wv, self.bv, self.log_std = map(f, (w1, b1, w2, b2, wa, ba, wv, bv, log_std))
        self.hidden, self.obs_dim = self.w1.shape
        self.act_dim = self.wa.shape[0]
        self._packed_key = None          # new tensors: the allocator may hand out an old address with a fresh version counter

    def _params(self):
        p = abi.PolicyParams()
        for name in ('w1', 'b1', 'w2', 'b2', 'wa', 'ba', 'wv', 'bv', 'log_std'):
            setattr(p, name, getattr(self, name).data_ptr())
        p.obs_dim, p.hidden, p.act_dim = self.obs_dim, self.hidden, self.act_dim
        return p

    def forward(self, obs, deterministic=False, eps=None, actions_out=None, values_out=None, log_probs_out=None):
        """obs: float32 cuda [N, obs_dim] (normalised).  eps: optional standard-normal draws [N, act_dim]; by default
        a counter-based stream keyed by (seed, call counter, global walker index).  The outputs may be
        rollout-buffer slots."""
        n = obs.shape[0]
        dev = obs.device
        a = torch.empty(n, self.act_dim, device=dev) if actions_out is None else actions_out
        v = torch.empty(n, device=dev) if values_out is None else values_out
        lp = torch.empty(n, device=dev) if log_probs_out is None else log_probs_out
        p = self._params()
        lib.check(self._lib.dl_policy_forward_packed(C.byref(p), _ptr(self._packed_weights()), _ptr(obs), n, _ptr(eps), self.seed, self.counter, self.index_base,
                                                     int(deterministic), _ptr(a), _ptr(v), _ptr(lp), _stream()))
        self.counter += 1
        return a, v, lp

    def torch_reference(self, obs, eps):
        """The same computation with torch ops (tests only)."""
        h = torch.tanh(torch.nn.functional.linear(obs, self.w1, self.b1))
        h = torch.tanh(torch.nn.functional.linear(h, self.w2, self.b2))
        mean = torch.nn.functional.linear(h, self.wa, self.ba)
        value = torch.nn.functional.linear(h, self.wv, self.bv)[:, 0]
        actions = mean + torch.exp(self.log_std) * eps
        logp = (-0.5 * eps ** 2 - self.log_std - 0.5 * math.log(2 * math.pi)).sum(1)
        return actions, value, logp
