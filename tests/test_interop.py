"""Python-boundary interop: `PPO(policy, env)` (drloco/train.py:110) runs SB3's
`_wrap_env`, which leaves an env alone only if `isinstance(env, VecEnv)`.  drloco_amd/compat.py derives HipVecEnv /
HipVecNormalize from SB3's own classes when stable-baselines3 imports.  The real packages are not installed here, so the
SB3 branch is exercised in a child interpreter against interface stubs (tests/stubs, see its README); the plain branch in
this interpreter."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, 'tests', 'stubs')


def run_with_stubs(code):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUBS, ROOT, os.environ.get('PYTHONPATH', '')]))
    p = subprocess.run([sys.executable, '-c', textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_plain_branch_without_sb3():
    from drloco_amd import compat
    if compat.HAVE_SB3:
        pytest.skip('stable-baselines3 is installed: the SB3 branch is the one in use')
    from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
    assert issubclass(HipVecNormalize, compat.VecEnvWrapperBase) and issubclass(HipVecEnv, compat.VecEnvBase)
    b = compat.make_box([-1, -2], [1, 2])
    assert b.shape == (2,) and b.dtype == np.float32 and b.contains(np.array([0.5, -1.5], np.float32)) and not b.contains(np.array([2, 0], np.float32))


def test_sb3_branch_class_protocol():
    """With stable_baselines3 / gym importable the classes ARE VecEnv / VecEnvWrapper subclasses with no abstract method
    left, and spaces are gym Boxes."""
    out = run_with_stubs('''
        import numpy as np
        from stable_baselines3.common.vec_env.base_vec_env import VecEnv, VecEnvWrapper
        import gym
        from drloco_amd import compat
        from drloco_amd.vec_env import HipVecEnv, HipVecNormalize
        assert compat.HAVE_SB3 and compat.GymBox is gym.spaces.Box
        assert issubclass(HipVecEnv, VecEnv) and issubclass(HipVecNormalize, VecEnvWrapper) and issubclass(HipVecNormalize, VecEnv)
        assert not getattr(HipVecEnv, '__abstractmethods__', None), HipVecEnv.__abstractmethods__
        assert not getattr(HipVecNormalize, '__abstractmethods__', None), HipVecNormalize.__abstractmethods__
        b = compat.make_box(np.full(29, -np.inf), np.full(29, np.inf))
        assert isinstance(b, gym.spaces.Box) and b.shape == (29,) and b.dtype == np.float32
        print('ok')
    ''')
    assert 'ok' in out


@pytest.mark.gpu
def test_sb3_wrap_env_takes_the_device_env_as_it_is():
    """_wrap_env's isinstance check, DummyVecEnv's space handling and the VecEnv round trip on a real device env."""
    out = run_with_stubs('''
        import numpy as np, gym
        from stable_baselines3.common.vec_env.base_vec_env import VecEnv, DummyVecEnv, wrap_env
        from drloco_amd.vec_env import vec_env, HipVecNormalize
        env = vec_env(num_envs=64, seed=3)                       # the reference's utils.vec_env signature
        assert isinstance(env, HipVecNormalize) and isinstance(env, VecEnv)
        wrapped = wrap_env(env)
        assert wrapped is env and not isinstance(wrapped, DummyVecEnv)
        for sp, dim in ((env.observation_space, 29), (env.action_space, 8)):
            assert isinstance(sp, gym.spaces.Box) and sp.shape == (dim,) and sp.dtype == np.float32
        assert (env.action_space.low == -300).all() and (env.action_space.high == 300).all()      # MujocoEnv: the actuators' ctrlrange
        obs = env.reset()
        assert obs.shape == (64, 29) and obs.dtype == np.float32
        obs, rew, done, infos = env.step(np.zeros((64, 8), np.float32))
        assert obs.shape == (64, 29) and rew.shape == (64,) and done.dtype == bool and len(infos) == 64
        assert len(env.get_attr('ep_len_smoothed')) == 64 and len(env.get_attr('moved_distance', indices=[3, 5])) == 2
        assert env.env_is_wrapped(type('Monitor', (), {})) == [True] * 64
        assert env.unwrapped is env.venv
        env.close()
        print('ok')
    ''')
    assert 'ok' in out
