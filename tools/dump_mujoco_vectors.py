#!/usr/bin/env python3
"""Dump input/output vectors of the REAL MuJoCo for the two walkers into tests/golden/G12_mujoco_step.npz.

The dynamics of the reference live in a third-party binary (MuJoCo via mujoco-py, drloco/mujoco/mimic_env.py:6-7,52,83)
that is neither in the reference checkout nor installable in the build container, so oracle/dl_oracle.c restates the
published pipeline and its parity with the binary is UNPINNED.  This script closes the gap on any machine
where `import mujoco` (the official bindings, MuJoCo >= 2.1.2) or `import mujoco_py` (what the reference uses) works:

    python tools/dump_mujoco_vectors.py --xml-dir /path/to/DRLoco/drloco/mujoco/xml        # writes tests/golden/G12_mujoco_step.npz

and the tests that consume the file (tests/test_oracle_golden.py::test_G12_*, tests/test_gpu_parity.py::test_G12_*) stop
skipping.  Per model (walker3d_flat_feet.xml, walker_165cm_65kg.xml; option overrides as the reference's env sets them:
none -- timestep 1 ms and RK4 come from the XML):
  fwd_*    256 random states (the generator of tests/test_gpu_parity.py::random_states, seed 12): (qpos, qvel, ctrl,
           qacc_warmstart) -> mj_forward -> qacc, ncon, nefc, qfrc_constraint
  roll_*   a 512-step zero-action rollout from qpos0 lifted by 5 cm (falls, lands, settles / topples): every step's
           (qpos, qvel, qacc_warmstart) BEFORE mj_step and (qpos, qvel, qacc_warmstart) AFTER it
  meta     MuJoCo version, body_invweight0 / dof_invweight0 / meaninertia of the compiled model (mj_setConst).
Nothing here is imported by the product or the tests; the npz is data."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = {'straight': ('walker3d_flat_feet.xml', 14, 8, (0.85, 1.3), 0.25), 'walker165': ('walker_165cm_65kg.xml', 19, 13, (0.75, 1.2), 0.2)}


class Official:
    """`mujoco` (DeepMind bindings)."""
    def __init__(self, xml):
        import mujoco
        self.mj = mujoco
        self.m = mujoco.MjModel.from_xml_path(xml)
        self.d = mujoco.MjData(self.m)
        self.version = 'mujoco ' + mujoco.__version__

    def set(self, q, v, u, w):
        self.mj.mj_resetData(self.m, self.d)
        self.d.qpos[:], self.d.qvel[:], self.d.ctrl[:], self.d.qacc_warmstart[:] = q, v, u, w

    def forward(self):
        self.mj.mj_forward(self.m, self.d)

    def step(self):
        self.mj.mj_step(self.m, self.d)

    def consts(self):
        return np.array(self.m.body_invweight0), np.array(self.m.dof_invweight0), float(self.m.stat.meaninertia)


class MujocoPy:
    """`mujoco_py` (MuJoCo 2.0 / 2.1, the reference's binding)."""
    def __init__(self, xml):
        import mujoco_py
        self.mp = mujoco_py
        self.model = mujoco_py.load_model_from_path(xml)
        self.sim = mujoco_py.MjSim(self.model)
        self.m, self.d = self.sim.model, self.sim.data
        self.version = 'mujoco_py ' + getattr(mujoco_py, '__version__', '?')

    def set(self, q, v, u, w):
        self.sim.reset()
        self.d.qpos[:], self.d.qvel[:], self.d.ctrl[:], self.d.qacc_warmstart[:] = q, v, u, w

    def forward(self):
        self.sim.forward()

    def step(self):
        self.sim.step()

    def consts(self):
        return np.array(self.m.body_invweight0), np.array(self.m.dof_invweight0), float(self.m.stat.meaninertia)


def backend(xml):
    try:
        return Official(xml)
    except ImportError:
        pass
    try:
        return MujocoPy(xml)
    except ImportError:
        raise SystemExit('neither `mujoco` nor `mujoco_py` imports on this machine: nothing to dump')


def random_states(qpos0, nv, nu, n, seed, zrange, spread):
    """tests/test_gpu_parity.py::random_states / test_loco3d_forward_dynamics (kept in step with them)."""
    rng = np.random.default_rng(seed)
    q = qpos0[:, None] + spread * rng.standard_normal((nv, n)); q[2] = rng.uniform(*zrange, n)
    v = 1.5 * rng.standard_normal((nv, n)); w = rng.standard_normal((nv, n)); u = rng.uniform(-300, 300, (nu, n))
    return q, v, w, u


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--xml-dir', default='/root/reference/drloco/mujoco/xml')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'G12_mujoco_step.npz'))
    ap.add_argument('--states', type=int, default=256)
    ap.add_argument('--steps', type=int, default=512)
    args = ap.parse_args()
    out = {}
    for key, (fname, nv, nu, zrange, spread) in MODELS.items():
        B = backend(os.path.join(args.xml_dir, fname))
        assert B.m.nv == nv and B.m.nu == nu and B.m.nq == nv, 'unexpected model sizes'
        qpos0 = np.array(B.m.qpos0)
        q, v, w, u = random_states(qpos0, nv, nu, args.states, 12, zrange, spread)
        qacc = np.zeros((nv, args.states)); fc = np.zeros_like(qacc); ncon = np.zeros(args.states, np.int32); nefc = np.zeros(args.states, np.int32)
        for i in range(args.states):
            B.set(q[:, i], v[:, i], u[:, i], w[:, i])
            B.forward()
            qacc[:, i], fc[:, i], ncon[i], nefc[i] = B.d.qacc, B.d.qfrc_constraint, B.d.ncon, B.d.nefc
        out.update({f'{key}__fwd_qpos': q, f'{key}__fwd_qvel': v, f'{key}__fwd_warm': w, f'{key}__fwd_ctrl': u, f'{key}__fwd_qacc': qacc,
                    f'{key}__fwd_qfrc_constraint': fc, f'{key}__fwd_ncon': ncon, f'{key}__fwd_nefc': nefc})
        q0 = qpos0.copy(); q0[2] += 0.05
        B.set(q0, np.zeros(nv), np.zeros(nu), np.zeros(nv))
        pre = np.zeros((args.steps, 3, nv)); post = np.zeros((args.steps, 3, nv))
        for t in range(args.steps):
            pre[t] = B.d.qpos, B.d.qvel, B.d.qacc_warmstart
            B.step()
            post[t] = B.d.qpos, B.d.qvel, B.d.qacc_warmstart
        out.update({f'{key}__roll_pre': pre, f'{key}__roll_post': post})
        bw, dw, mi = B.consts()
        out.update({f'{key}__body_invweight0': bw, f'{key}__dof_invweight0': dw, f'{key}__meaninertia': np.float64(mi),
                    f'{key}__timestep': np.float64(B.m.opt.timestep), f'{key}__integrator': np.int32(B.m.opt.integrator)})
        out['version'] = np.array(B.version)
    np.savez_compressed(args.out, **out)
    print('wrote', args.out, 'from', out['version'])


if __name__ == '__main__':
    main()
