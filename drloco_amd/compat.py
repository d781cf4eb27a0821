"""Object-protocol interop with the reference's callers: stable-baselines3's VecEnv classes and gym's Box.

`PPO(policy, env, ...)` (drloco/train.py:110) runs `BaseAlgorithm._wrap_env`, which wraps anything that is not an
`isinstance(env, VecEnv)` into a `DummyVecEnv`, and SB3 reads `observation_space` / `action_space` as `gym.spaces.Box`.
When those packages import, HipVecEnv / HipVecNormalize therefore derive from SB3's own `VecEnv` / `VecEnvWrapper` and carry
real `gym.spaces.Box` spaces; when they do not (this image: neither is installed) the same classes stand on plain `object`
and the minimal Box below.  Nothing else of either package is used.  tests/test_interop.py exercises both branches (the SB3
branch against stub modules that restate the 1.0 interface)."""
import numpy as np


def _find_sb3():
    try:
        from stable_baselines3.common.vec_env.base_vec_env import VecEnv, VecEnvWrapper
        return VecEnv, VecEnvWrapper
    except Exception:            # not installed, or an installation that does not import on this interpreter
        return None, None


def _find_box():
    for mod in ('gym.spaces', 'gymnasium.spaces'):        # SB3 1.0 uses gym; later releases gymnasium
        try:
            return __import__(mod, fromlist=['Box']).Box
        except Exception:
            continue
    return None


SB3_VecEnv, SB3_VecEnvWrapper = _find_sb3()
GymBox = _find_box()
HAVE_SB3 = SB3_VecEnv is not None


class MiniBox:
    """Minimal stand-in for gym.spaces.Box (shape/low/high/dtype/sample/contains) when gym is absent."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        shape = np.shape(low) if shape is None else tuple(shape)
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'


def make_box(low, high, dtype=np.float32):
    """gym.spaces.Box(low, high) from per-component bounds (arrays of equal shape) if gym imports, else the stand-in."""
    low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
    if GymBox is not None:
        return GymBox(low=low, high=high, dtype=dtype)
    return MiniBox(low, high, low.shape, dtype)


class _PlainVecEnv:
    """What HipVecEnv needs from SB3's VecEnv when SB3 is absent: the constructor's three attributes."""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


class _PlainVecEnvWrapper(_PlainVecEnv):
    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        _PlainVecEnv.__init__(self, venv.num_envs, observation_space or venv.observation_space, action_space or venv.action_space)


VecEnvBase = SB3_VecEnv if HAVE_SB3 else _PlainVecEnv
VecEnvWrapperBase = SB3_VecEnvWrapper if HAVE_SB3 else _PlainVecEnvWrapper
