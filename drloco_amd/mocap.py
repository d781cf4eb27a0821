"""Mocap ingestion: the reference's step-segmented MATLAB file -> flat table.

Restates StraightWalkingTrajectories' load-time processing
(drloco/ref_trajecs/straight_walk_trajecs.py):
  _load_ref_trajecs          :304-320   data['Data'].flatten() -> per-step (rows, len) arrays
  _calculate_walking_speed   :393-415   mean COM-x velocity per step, exp. smoothing alpha = 0.2
                                        (smooth_exponential, drloco/common/utils.py:264-268)
  _determine_left_steps_indices :221-230 max knee velocity L > R
and the walker's row selection (drloco/mujoco/mimic_walker3d.py:11-23).
"""
import os

import numpy as np

from . import abi

# row indices of the two file layouts (straight_walk_trajecs.py:59-91): the trunk's Euler angles are rows 35-37 of the 38-row
# constant-speed file and rows 37-39 of the 40-row speed-ramp file (the reference's default PATH_REF_TRAJECS, :22-27), where the two
# ground-reaction-force rows sit at 35-36; everything else is shared
_TRUNK_EULER_ROW0 = {38: 35, 40: 37}
_QVEL_ROWS = [15, 16, 17, 18, 19, 20, 22, 21, 23, 24, 26, 25, 27, 28]
_KNEE_VEL_R, _KNEE_VEL_L, _COM_VEL_X = 23, 27, 15


def _qpos_rows(n_rows):
    e = _TRUNK_EULER_ROW0[n_rows]
    return [0, 1, 2, e, e + 1, e + 2, 8, 7, 9, 10, 12, 11, 13, 14]

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
DEFAULT_TABLE = os.path.join(DATA_DIR, 'straight_walk_const400.npz')


class RefTable:
    """Flat reference table: table[2*nv, total_len] (qpos rows then qvel rows, model order)."""

    def __init__(self, table, step_off, step_is_left, step_vel, stride):
        self.table = np.ascontiguousarray(table, dtype=np.float64)
        self.step_off = np.ascontiguousarray(step_off, dtype=np.int32)
        self.step_is_left = np.ascontiguousarray(step_is_left, dtype=np.int32)
        self.step_vel = np.ascontiguousarray(step_vel, dtype=np.float64)
        self.stride = int(stride)
        assert self.table.shape[1] == self.step_off[-1]

    @property
    def n_steps(self):
        return len(self.step_off) - 1

    @property
    def step_len(self):
        return np.diff(self.step_off)

    def as_desc(self):
        import ctypes as C
        d = abi.RefsDesc()
        d.n_steps, d.n_rows, d.total_len, d.stride = self.n_steps, self.table.shape[0], self.table.shape[1], self.stride
        d.table = self.table.ctypes.data_as(C.POINTER(C.c_double))
        d.step_off = self.step_off.ctypes.data_as(C.POINTER(C.c_int32))
        d.step_is_left = self.step_is_left.ctypes.data_as(C.POINTER(C.c_int32))
        d.step_vel = self.step_vel.ctypes.data_as(C.POINTER(C.c_double))
        d._keepalive = self
        return d

    def mirrored(self):
        """StraightWalkingTrajectories._mirror_refs (straight_walk_trajecs.py:72-94,128-139) on the converted table:
        every left step becomes the preceding right step with the right/left leg rows exchanged and the lateral
        quantities negated (COM y, trunk rotation about x and z, frontal hip angle -- positions and velocities).  In
        the straight walker's row order (mimic_walker3d.py:11-23): qpos = [com x y z, trunk rot x y z,
        hip sagittal/frontal, knee, ankle (right), the same (left)], velocities likewise.  Step velocities and the
        left-step flags keep the values computed from the recorded data (the reference mirrors after computing them)."""
        if self.table.shape[0] != 28:
            raise ValueError('mirroring is defined for the straight walker\'s step-segmented table')
        half = list(range(6)) + list(range(10, 14)) + list(range(6, 10))
        perm = half + [14 + r for r in half]
        neg = [1, 3, 5, 7, 11]
        neg = neg + [14 + r for r in neg]
        steps = [self.table[:, self.step_off[i]:self.step_off[i + 1]] for i in range(self.n_steps)]
        for i in np.nonzero(self.step_is_left)[0]:
            m = steps[i - 1][perm, :].copy()
            m[neg, :] *= -1
            steps[i] = m
        off = np.concatenate([[0], np.cumsum([s.shape[1] for s in steps])])
        return RefTable(np.concatenate(steps, axis=1), off, self.step_is_left, self.step_vel, self.stride)

    def save(self, path):
        np.savez_compressed(path, table=self.table, step_off=self.step_off, step_is_left=self.step_is_left,
                            step_vel=self.step_vel, stride=np.int32(self.stride))

    @classmethod
    def load(cls, path=DEFAULT_TABLE):
        z = np.load(path)
        return cls(z['table'], z['step_off'], z['step_is_left'], z['step_vel'], int(z['stride']))


def hip3d_qpos(qpos):
    """StraightWalking3dHipTrajectories.get_qpos (straight_walk_hip3d_trajecs.py:8-12): the straight walker's 14 reference
    positions with a constant frontal ("traversal") hip angle inserted per leg, -0.05 rad before the right knee row group and
    +0.05 rad for the left leg -> 16 values.  No environment of the reference's env_map uses the class (a 16-dof walker is not in
    the repository), so this is a data transformation only; pinned by golden G13."""
    q = np.asarray(qpos, dtype=np.float64)
    return np.concatenate([q[..., :8], np.full(q.shape[:-1] + (1,), -0.05), q[..., 8:12], np.full(q.shape[:-1] + (1,), 0.05), q[..., 12:]], axis=-1)


def hip3d_qvel(qvel):
    """StraightWalking3dHipTrajectories.get_qvel (straight_walk_hip3d_trajecs.py:14-19): zero velocity for the two added hip rows."""
    v = np.asarray(qvel, dtype=np.float64)
    z = np.zeros(v.shape[:-1] + (1,))
    return np.concatenate([v[..., :8], z, v[..., 8:12], z, v[..., 12:]], axis=-1)


def _sequential_mean(row):
    acc = 0.0
    for x in row.tolist():
        acc = acc + x
    return acc / len(row)


def convert_straight_walk_mat(mat_path, sample_freq=400, control_freq=200, mirror_refs=False):
    """Trajecs_Constant_Speed_400Hz.mat (38 rows per step) or Trajecs_Ramp_Slow_400Hz_EulerTrunkAdded.mat (40 rows per step: the
    reference's default file, straight_walk_trajecs.py:22-27; any number of steps) -> RefTable.  mirror_refs:
    StraightWalkingTrajectories(mirror_refs=True) (straight_walk_trajecs.py:98-116): applied after the step velocities and the
    left-step indices were computed from the recorded data, exactly as the reference orders it."""
    import scipy.io as spio
    raw = spio.loadmat(mat_path, squeeze_me=True)['Data'].flatten()
    n_rows = raw[0].shape[0]
    if n_rows not in _TRUNK_EULER_ROW0 or any(s.shape[0] != n_rows for s in raw):
        raise ValueError(f'{mat_path}: {n_rows} rows per step; the reference defines the 38-row (constant speed) and the 40-row (speed ramp, GRF rows 35-36) layouts')
    stride = sample_freq / control_freq
    if stride != int(stride):
        raise ValueError('sample frequency must be an integer multiple of the control frequency')
    steps = [np.asarray(s, dtype=np.float64) for s in raw]
    lens = [s.shape[1] for s in steps]
    off = np.concatenate([[0], np.cumsum(lens)])
    table = np.concatenate([s[_qpos_rows(n_rows) + _QVEL_ROWS, :] for s in steps], axis=1)
    is_left = [int(np.max(s[_KNEE_VEL_L]) > np.max(s[_KNEE_VEL_R])) for s in steps]
    # _calculate_walking_speed (:393-415) calls np.mean on the rows as loadmat returns them: the reference's files hold every sample as a 1 x 1
    # cell, i.e. dtype=object arrays, whose mean adds left to right; a file with plain float matrices gets numpy's pairwise sum
    vel = np.array([_sequential_mean(s[_COM_VEL_X]) if r.dtype == object else float(np.mean(s[_COM_VEL_X])) for r, s in zip(raw, steps)])
    for t in range(1, len(vel)):
        vel[t] = 0.2 * vel[t] + 0.8 * vel[t - 1]
    for s in steps:
        if not s[0, 0] < 0.005:
            raise ValueError('COM-x of every step must start at 0 (straight_walk_trajecs.py:343)')
    ref = RefTable(table, off, is_left, vel, int(stride))
    return ref.mirrored() if mirror_refs else ref


def synthetic_straight_walk(n_steps=250, seed=0, n_rows=40, len_lo=56, len_hi=84, sample_freq=400.0):
    """Synthetic stand-in for the missing Trajecs_Ramp_Slow_400Hz_EulerTrunkAdded.mat (.MISSING_LARGE_BLOBS:2) in the reference's
    step-segmented schema: a list of n_steps float64 arrays (n_rows, len_i) with the row meaning of straight_walk_trajecs.py:29-91 --
    COM-x starting at 0 in every step, a walking speed that ramps over the steps (0.7 -> 1.3 m/s with noise), alternating swing legs
    (the swing knee is the faster one), distinct values in the GRF rows (35-36) and the trunk's Euler rows (37-39) so that a wrong row
    map shows.  Steps are shorter than real ones (56-84 samples instead of ~260) to keep the fixture small."""
    rng = np.random.default_rng(seed)
    steps = []
    for i in range(n_steps):
        L = int(rng.integers(len_lo, len_hi + 1))
        t = np.arange(L) / sample_freq
        u = np.arange(L) / L
        s = np.zeros((n_rows, L))
        speed = 0.7 + 0.6 * i / max(1, n_steps - 1) + 0.05 * rng.standard_normal()
        left = (i % 2 == 1) != (i % 17 == 16)       # mostly alternating; now and then a side repeats (the list is data, not parity); step 0 is a right step
        vx = speed * (1 + 0.1 * np.sin(2 * np.pi * u + rng.uniform(0, 6.28)))
        s[15] = vx; s[0] = np.concatenate([[0.0], np.cumsum(vx[:-1]) / sample_freq])
        s[1] = 0.03 * np.sin(2 * np.pi * u) * (1 if left else -1); s[16] = 0.03 * 2 * np.pi / (L / sample_freq) * np.cos(2 * np.pi * u) * (1 if left else -1)
        s[2] = 1.05 + 0.02 * np.cos(4 * np.pi * u); s[17] = -0.02 * 4 * np.pi / (L / sample_freq) * np.sin(4 * np.pi * u)
        q = rng.standard_normal(4) * 0.02 + np.array([1, 0, 0, 0]); s[3:7] = (q / np.linalg.norm(q))[:, None]
        for r in range(7, 15):
            a, ph = rng.uniform(0.05, 0.4), rng.uniform(0, 6.28)
            s[r] = a * np.sin(2 * np.pi * u + ph); s[r + 14] = a * 2 * np.pi / (L / sample_freq) * np.cos(2 * np.pi * u + ph)
        s[18:21] = 0.2 * rng.standard_normal((3, 1)) * np.cos(2 * np.pi * u)[None, :]
        # the swing knee moves faster: the left-step list is computed from these two rows (:221-230)
        s[_KNEE_VEL_R] *= 0.5; s[_KNEE_VEL_L] *= 0.5
        s[_KNEE_VEL_L if left else _KNEE_VEL_R] += 3.0 * np.sin(np.pi * u)
        s[29:35] = 0.1 * rng.standard_normal((6, 1)) + 0.01 * np.sin(2 * np.pi * u)[None, :]
        e = _TRUNK_EULER_ROW0[n_rows]
        if n_rows == 40:
            s[35] = 800.0 * np.sin(np.pi * u) * (0 if left else 1); s[36] = 800.0 * np.sin(np.pi * u) * (1 if left else 0)
        s[e:e + 3] = 0.05 * rng.standard_normal((3, 1)) + 0.03 * np.sin(2 * np.pi * u + 1.0)[None, :] * np.array([[1.0], [0.5], [-0.7]])
        steps.append(s)
    return steps


def write_straight_walk_mat(path, steps, nested=True):
    """Write step arrays in the reference's file schema: 'Data' = a 1 x n MATLAB cell array of steps.  nested=True stores every sample as
    a 1 x 1 cell, as the reference's own files do (loadmat then returns dtype=object arrays and numpy reductions on them add left to
    right); nested=False stores plain double matrices."""
    import scipy.io as spio
    data = np.empty((1, len(steps)), dtype=object)
    for i, s in enumerate(steps):
        if nested:
            o = np.empty(s.shape, dtype=object)
            o[...] = s
            data[0, i] = o
        else:
            data[0, i] = np.asarray(s, np.float64)
    spio.savemat(path, {'Data': data}, do_compression=True)


# loco3d: rows of angJoi / angDJoi used by MimicWalker165cm65kg (mimic_walker_165cm_65kg.py:6-15,
# loco3d_trajecs.py:7-18): pelvis tx, tz, ty, list, tilt, rotation, lumbar bending, extension,
# rotation, right hip flexion/adduction/rotation, knee, ankle, left ditto
LOCO3D_ROWS = [3, 5, 4, 1, 0, 2, 21, 20, 22, 6, 7, 8, 9, 10, 13, 14, 15, 16, 17]


def convert_loco3d_mat(mat_path, sample_freq=500, control_freq=100, adaptations=None):
    """loco3d_guoping.mat (angJoi, angDJoi: (37, L)) -> RefTable with one continuous trajectory."""
    import scipy.io as spio
    d = spio.loadmat(mat_path, squeeze_me=True)
    return loco3d_table(np.asarray(d['angJoi'], np.float64), np.asarray(d['angDJoi'], np.float64), sample_freq, control_freq, adaptations)


def loco3d_table(ang, ang_vel, sample_freq=500, control_freq=100, adaptations=None):
    """adaptations: BaseReferenceTrajectories.adapt_trajectories (base_ref_trajecs.py:105-118) -- {row of angJoi/angDJoi:
    scalar}; position and velocity of that row are scaled (the reference's walkers pass an empty dict)."""
    if adaptations:
        ang, ang_vel = np.array(ang, np.float64), np.array(ang_vel, np.float64)
        for row, scalar in adaptations.items():
            ang[row, :] *= scalar
            ang_vel[row, :] *= scalar
    stride = sample_freq / control_freq
    if stride != int(stride):
        raise ValueError('sample frequency must be an integer multiple of the control frequency')
    table = np.concatenate([ang[LOCO3D_ROWS, :], ang_vel[LOCO3D_ROWS, :]], axis=0)
    L = table.shape[1]
    return RefTable(table, [0, L], [0], [0.0], int(stride))


def synthetic_loco3d(L=60000, seed=0, sample_freq=500.0):
    """Synthetic stand-in for the missing loco3d_guoping.mat (a large blob absent from the reference
    checkout, .MISSING_LARGE_BLOBS:1) with the same schema: angJoi / angDJoi (37, L) float64 at 500 Hz,
    8 clips of band-limited (<= 3 Hz) sinusoid sums scaled to the joint ranges of
    walker_165cm_65kg.xml, pelvis translating forward at ~1.2 m/s."""
    rng = np.random.default_rng(seed)
    t = np.arange(L) / sample_freq
    ang = np.zeros((37, L)); vel = np.zeros((37, L))
    amp = {0: 0.05, 1: 0.04, 2: 0.06, 21: 0.05, 20: 0.05, 22: 0.05, 6: 0.45, 7: 0.08, 8: 0.06, 9: 0.5, 10: 0.2,
           13: 0.45, 14: 0.08, 15: 0.06, 16: 0.5, 17: 0.2}
    mid = {9: -0.6, 16: -0.6, 7: -0.05, 14: -0.05}
    clip = (np.arange(L) * 8) // L
    for row in range(37):
        a = amp.get(row, 0.02)
        for c in range(8):
            m = clip == c
            f = rng.uniform(0.6, 3.0, 3); ph = rng.uniform(0, 2 * np.pi, 3); w = rng.uniform(0.2, 1.0, 3); w /= w.sum()
            for k in range(3):
                ang[row, m] += a * w[k] * np.sin(2 * np.pi * f[k] * t[m] + ph[k])
                vel[row, m] += a * w[k] * 2 * np.pi * f[k] * np.cos(2 * np.pi * f[k] * t[m] + ph[k])
        ang[row] += mid.get(row, 0.0)
    # pelvis translation: tx forward, tz (sim -y) small sway, ty height
    ang[3] = 1.2 * t + 0.02 * np.sin(2 * np.pi * 1.8 * t); vel[3] = 1.2 + 0.02 * 2 * np.pi * 1.8 * np.cos(2 * np.pi * 1.8 * t)
    ang[5] = 0.03 * np.sin(2 * np.pi * 0.9 * t); vel[5] = 0.03 * 2 * np.pi * 0.9 * np.cos(2 * np.pi * 0.9 * t)
    ang[4] = 0.95 + 0.02 * np.sin(2 * np.pi * 1.8 * t + 0.4); vel[4] = 0.02 * 2 * np.pi * 1.8 * np.cos(2 * np.pi * 1.8 * t + 0.4)
    return ang, vel


if __name__ == '__main__':
    import sys
    src = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/mocaps/straight_walking/Trajecs_Constant_Speed_400Hz.mat'
    convert_straight_walk_mat(src).save(DEFAULT_TABLE)
    print('wrote', DEFAULT_TABLE)
