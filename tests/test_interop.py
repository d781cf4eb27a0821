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


def test_model_zip_round_trips_through_sb3s_reader(tmp_path):
    """drloco_amd.checkpoint.write_model_zip (the reference's model.save, drloco/common/utils.py:175-192) in SB3 1.0's save_to_zip_file
    layout: written HERE (no gym / SB3 / drloco importable: the class paths inside come from placeholders), read in a child interpreter
    by the `load_from_zip_file`-shaped reader of tests/stubs with those paths importable.  Unpinned by a real SB3 archive."""
    import types
    import zipfile
    import torch
    from drloco_amd import checkpoint as ck
    g = torch.Generator().manual_seed(3)
    t = lambda *sh: torch.randn(*sh, generator=g).requires_grad_()
    pol = types.SimpleNamespace(w1=t(64, 29), b1=t(64), w2=t(64, 64), b2=t(64), wa=t(8, 64), ba=t(8), wv=t(1, 64), bv=t(1), log_std=torch.full((8,), -0.75).requires_grad_())
    names = ('wv', 'log_std', 'w1', 'b1', 'w2', 'b2', 'wa', 'ba', 'bv')                       # the caller's own order: matched by identity
    opt = torch.optim.Adam([getattr(pol, k) for k in names], lr=3e-4, eps=1e-5)
    sum((getattr(pol, k) ** 2).sum() for k in names).backward()
    opt.step()
    z = str(tmp_path / 'model_5.zip')
    ck.write_model_zip(pol, z, optimizer=opt, hyper=dict(n_envs=128, num_timesteps=81920))
    assert {'data', 'policy.pth', 'policy.optimizer.pth', 'pytorch_variables.pth', '_stable_baselines3_version'} == set(zipfile.ZipFile(z).namelist())
    assert not any(m.split('.')[0] in ('gym', 'drloco') for m in sys.modules), 'placeholder modules must not stay behind'
    back = ck.read_policy_zip(z)                                                              # our own reader still takes it
    assert torch.equal(back['w2'], pol.w2.detach())
    out = run_with_stubs(f'''
        import gym, numpy as np, torch
        from stable_baselines3.common.save_util import load_from_zip_file
        data, params, pv = load_from_zip_file({z!r})
        ob, ac = data['observation_space'], data['action_space']
        assert isinstance(ob, gym.spaces.Box) and ob.shape == (29,) and ob.dtype == np.float32 and np.isinf(ob.low).all()
        assert isinstance(ac, gym.spaces.Box) and ac.shape == (8,) and (ac.low == -300).all() and (ac.high == 300).all()
        from drloco.custom.policies import CustomActorCriticPolicy
        assert data['policy_class'] is CustomActorCriticPolicy and abs(data['policy_kwargs']['log_std_init'] + 0.75) < 1e-3
        assert data['gamma'] == 0.995 and data['n_envs'] == 128 and data['num_timesteps'] == 81920 and data['clip_range'] == 0.15
        assert data['clip_range_vf'] == 0.15 and data['n_steps'] == 2048 and data['batch_size'] == 2048 and (data['lr_start'], data['lr_final']) == (5e-4, 1e-6)      # train.py:110-118
        sd = params['policy']
        order = ['log_std', 'mlp_extractor.policy_net.0.weight', 'mlp_extractor.policy_net.0.bias', 'mlp_extractor.policy_net.2.weight', 'mlp_extractor.policy_net.2.bias',
                 'action_net.weight', 'action_net.bias', 'value_net.weight', 'value_net.bias']
        assert set(order) <= set(sd) and torch.equal(sd['mlp_extractor.value_net.2.weight'], sd['mlp_extractor.policy_net.2.weight'])
        # policy.optimizer.pth loads into Adam over the parameters in ActorCriticPolicy.parameters() order, moments attached to the right tensors
        ps = [torch.nn.Parameter(sd[k].clone()) for k in order]
        opt = torch.optim.Adam(ps, lr=1.0, eps=1e-5)
        opt.load_state_dict(params['policy.optimizer'])
        assert opt.param_groups[0]['lr'] == 3e-4 and len(opt.state) == 9
        for p in ps:
            st = opt.state[p]
            assert st['exp_avg'].shape == p.shape and float(st['step']) == 1
            assert torch.allclose(st['exp_avg'], 0.1 * 2 * p.detach(), rtol=1e-4, atol=1e-4)          # (1 - beta1) x the gradient 2 p of sum(p^2), p one step of 3e-4 before the saved weights
        assert pv == {{}}
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
        assert env.env_is_wrapped(type('Monitor', (), {'__module__': 'drloco.mujoco.monitor_wrapper'})) == [True] * 64
        assert env.env_is_wrapped(type('Monitor', (), {})) == [False] * 64          # a class that is merely CALLED Monitor is not one
        assert env.unwrapped is env.venv
        env.close()
        print('ok')
    ''')
    assert 'ok' in out
