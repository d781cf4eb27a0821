"""Interface stub (see tests/stubs/README.md): the abstract surface of SB3 1.0's VecEnv / VecEnvWrapper."""
import inspect
from abc import ABC, abstractmethod


class VecEnv(ABC):
    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def step_async(self, actions): ...

    @abstractmethod
    def step_wait(self): ...

    @abstractmethod
    def close(self): ...

    @abstractmethod
    def get_attr(self, attr_name, indices=None): ...

    @abstractmethod
    def set_attr(self, attr_name, value, indices=None): ...

    @abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...

    @abstractmethod
    def env_is_wrapped(self, wrapper_class, indices=None): ...

    @abstractmethod
    def seed(self, seed=None): ...

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        raise NotImplementedError

    def render(self, mode='human'):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    def _get_indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else indices


class VecEnvWrapper(VecEnv):
    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        VecEnv.__init__(self, num_envs=venv.num_envs, observation_space=observation_space or venv.observation_space,
                        action_space=action_space or venv.action_space)
        self.class_attributes = dict(inspect.getmembers(self.__class__))

    def step_async(self, actions):
        self.venv.step_async(actions)

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def step_wait(self): ...

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def close(self):
        return self.venv.close()

    def get_attr(self, attr_name, indices=None):
        return self.venv.get_attr(attr_name, indices)

    def set_attr(self, attr_name, value, indices=None):
        return self.venv.set_attr(attr_name, value, indices)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return self.venv.env_method(method_name, *method_args, indices=indices, **method_kwargs)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return self.venv.env_is_wrapped(wrapper_class, indices=indices)


class DummyVecEnv(VecEnv):
    """What _wrap_env would put around a non-VecEnv: here only a marker that the wrap happened."""

    def __init__(self, env_fns):
        self.envs = [f() for f in env_fns]
        VecEnv.__init__(self, len(env_fns), self.envs[0].observation_space, self.envs[0].action_space)

    reset = step_async = step_wait = close = get_attr = set_attr = env_method = env_is_wrapped = seed = lambda self, *a, **k: None


def wrap_env(env):
    """BaseAlgorithm._wrap_env (SB3 1.0), the part that matters here: anything that is not a VecEnv gets a DummyVecEnv."""
    if not isinstance(env, VecEnv):
        env = DummyVecEnv([lambda: env])
    return env
