#!/usr/bin/env python3
"""Condense a checkpoint of examples/train_ppo.py --save (models/model_<k>.zip in SB3 1.0's layout + envs/env_<k>, the VecNormalize pickle) into the small
file the contact-rich benchmark line loads: drloco_amd/data/walking_policy.npz = the nine policy tensors (float32, 282 641 parameters) + VecNormalize's
moments + how it was trained.  usage: python tools/pack_walking_ckpt.py gpurun_out/ckpt/s2 80 [out.npz] [note]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drloco_amd import checkpoint


def main():
    src, ckpt = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'drloco_amd', 'data', 'walking_policy.npz')
    note = sys.argv[4] if len(sys.argv) > 4 else ''
    t = checkpoint.read_policy_zip(os.path.join(src, 'models', f'model_{ckpt}.zip'))
    vn = checkpoint.read_vecnormalize(os.path.join(src, 'envs', f'env_{ckpt}'))
    meta = dict(trained_by='examples/train_ppo.py --mio 8 (PPO, the reference\'s hyperparameters: drloco/config/hypers.py; 128 walkers x 128-step rollouts, exact per-step moments)',
                env='straight walker (walker3d_flat_feet.xml), packaged 30-step constant-speed mocap table, policy mirroring on', note=note,
                clip_obs=vn['clip_obs'], clip_reward=vn['clip_reward'], gamma=vn['gamma'], epsilon=vn['epsilon'])
    np.savez_compressed(out, **{k: v.numpy().astype(np.float32) for k, v in t.items()},
                        obs_mean=vn['obs_rms']['mean'], obs_var=vn['obs_rms']['var'], obs_count=np.float64(vn['obs_rms']['count']),
                        ret_mean=np.float64(vn['ret_rms']['mean']), ret_var=np.float64(vn['ret_rms']['var']), ret_count=np.float64(vn['ret_rms']['count']),
                        meta=np.array(json.dumps(meta)))
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
