"""ctypes mirror of include/drloco_hip.h (POD descriptors + constants).

Kept byte-compatible with the C header; tests/test_abi.py checks sizeof() against the
library's own view (dl_abi_sizeof) and that every declared symbol is exported.
"""
import ctypes as C

DL_ABI_VERSION = 7
DL_ADV_WORKSPACE_BYTES = (2 * 512 + 2) * 8


def vn_workspace_bytes(obs_dim):
    """DL_VN_WORKSPACE_BYTES(D) of include/drloco_hip.h"""
    return 8 * (2 * 32 * (obs_dim + 1) + 2)


def vn_steps_workspace_bytes(k, b, obs_dim):
    """DL_VN_STEPS_WORKSPACE_BYTES(K, B, D) of include/drloco_hip.h"""
    return 8 * (k * b + k * 32 * (obs_dim + 1) * 2 + k * (obs_dim + 1) * 2)

DL_MAX_BODY, DL_MAX_DOF, DL_MAX_GEOM, DL_MAX_SITE, DL_MAX_ACT = 12, 20, 12, 8, 16
DL_JNT_SLIDE, DL_JNT_HINGE = 0, 1
DL_ENV_STRAIGHT, DL_ENV_LOCO3D = 0, 1
DL_GEOM_CAPSULE, DL_GEOM_BOX = 0, 1
(DL_CUR_I_STEP, DL_CUR_POS, DL_CUR_RSI_STEP, DL_CUR_COUNT, DL_CUR_EP_DUR, DL_CUR_HAS_DIST,
 DL_CUR_EPISODE, DL_CUR_READ_STEP, DL_CUR_EVAL_K) = range(9)
DL_CUR_WORDS = 9
EVAL_N_TIMES = 20        # drloco/config/config.py:23
DL_ROLLOUT_PERSISTENT, DL_ROLLOUT_MOMENTS_PER_ROLLOUT, DL_ROLLOUT_WORKGROUP_TILES, DL_ROLLOUT_DETERMINISTIC = 1, 2, 4, 8
DL_OK, DL_E_INVAL, DL_E_NODEVICE, DL_E_HIP, DL_E_NOMEM, DL_E_FAULT = 0, -1, -2, -3, -4, -5
DL_FAULT_DYN_TIMEOUT, DL_FAULT_SRV_TIMEOUT, DL_FAULT_GRID_TIMEOUT = 1, 2, 4          # bits of the fault word (dl_fault_check)
DL_INTENDED_COUNT_PER_EPISODE, DL_INTENDED_EVAL_OWN_STEP, DL_INTENDED_COMZ_PER_EPISODE = 2, 4, 8          # bits of dl_config.intended_semantics (quirks Q2, Q3, Q4 off)
DL_INTENDED_ALL = 2 | 4 | 8

_d, _i = C.c_double, C.c_int32


class ModelDesc(C.Structure):
    _fields_ = [
        ('nbody', _i), ('nv', _i), ('nu', _i), ('ngeom', _i), ('nsite', _i), ('frame_skip', _i),
        ('timestep', _d), ('gravity', _d * 3), ('solref', _d * 2), ('solimp', _d * 5),
        ('tolerance', _d), ('ls_tolerance', _d), ('iterations', _i), ('ls_iterations', _i),
        ('body_parent', _i * DL_MAX_BODY), ('body_pos', (_d * 3) * DL_MAX_BODY),
        ('body_mass', _d * DL_MAX_BODY), ('body_ipos', (_d * 3) * DL_MAX_BODY),
        ('body_inertia', (_d * 3) * DL_MAX_BODY),
        ('jnt_type', _i * DL_MAX_DOF), ('jnt_body', _i * DL_MAX_DOF),
        ('jnt_axis', (_d * 3) * DL_MAX_DOF), ('jnt_pos', (_d * 3) * DL_MAX_DOF),
        ('jnt_qpos0', _d * DL_MAX_DOF), ('jnt_limited', _i * DL_MAX_DOF),
        ('jnt_range', (_d * 2) * DL_MAX_DOF), ('jnt_damping', _d * DL_MAX_DOF),
        ('jnt_armature', _d * DL_MAX_DOF),
        ('geom_type', _i * DL_MAX_GEOM), ('geom_body', _i * DL_MAX_GEOM),
        ('geom_pos', (_d * 3) * DL_MAX_GEOM), ('geom_mat', (_d * 9) * DL_MAX_GEOM),
        ('geom_size', (_d * 3) * DL_MAX_GEOM), ('geom_friction', _d * DL_MAX_GEOM),
        ('floor_friction', _d),
        ('site_body', _i * DL_MAX_SITE), ('site_pos', (_d * 3) * DL_MAX_SITE),
        ('act_dof', _i * DL_MAX_ACT), ('act_gear', _d * DL_MAX_ACT),
        ('act_ctrlrange', (_d * 2) * DL_MAX_ACT), ('act_forcerange', (_d * 2) * DL_MAX_ACT),
        ('body_invweight0', (_d * 2) * DL_MAX_BODY), ('dof_invweight0', _d * DL_MAX_DOF),
        ('meaninertia', _d),
    ]


class RefsDesc(C.Structure):
    _fields_ = [
        ('n_steps', _i), ('n_rows', _i), ('total_len', _i), ('stride', _i),
        ('table', C.POINTER(_d)), ('step_off', C.POINTER(_i)), ('step_is_left', C.POINTER(_i)),
        ('step_vel', C.POINTER(_d)),
    ]


class Config(C.Structure):
    _fields_ = [
        ('rew_weights', _d * 3), ('rew_scale', _d), ('alive_bonus', _d), ('com_z_min', _d),
        ('ctrl_freq', _d), ('ep_dur_max', _i), ('mirror_policy', _i), ('precision', _i),
        ('env_index_base', _i), ('seed', C.c_uint64), ('env_kind', _i), ('lanes_per_walker', _i),
        ('intended_semantics', _i), ('strict_solver', _i),
    ]


class PolicyParams(C.Structure):
    _fields_ = [('w1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('wa', C.c_void_p), ('ba', C.c_void_p),
                ('wv', C.c_void_p), ('bv', C.c_void_p), ('log_std', C.c_void_p), ('obs_dim', _i), ('hidden', _i), ('act_dim', _i)]


class VecNormState(C.Structure):
    """dl_vecnorm_state (include/drloco_hip.h)."""
    _fields_ = [('obs_mean', C.c_void_p), ('obs_var', C.c_void_p), ('obs_count', C.c_void_p), ('ret', C.c_void_p), ('ret_mean', C.c_void_p),
                ('ret_var', C.c_void_p), ('ret_count', C.c_void_p), ('workspace', C.c_void_p), ('gamma', C.c_double), ('eps', C.c_double),
                ('clip_obs', C.c_double), ('clip_rew', C.c_double), ('flags', _i)]


def loco3d_config(**kw):
    """Defaults for MimicWalker165cm65kg: CTRL_FREQ 100 (config.py:20); the policy-mirroring modification
    must be off because Loco3dReferenceTrajectories has no is_step_left."""
    base = dict(env_kind=DL_ENV_LOCO3D, ctrl_freq=100.0, mirror_policy=0)
    base.update(kw)
    return default_config(**base)


def default_config(**kw):
    """Constants of drloco/config/hypers.py:48-58 and config.py:20 (reference defaults)."""
    c = Config()
    c.rew_weights[:] = [0.8, 0.2, 0.0]
    c.rew_scale = 1.0
    c.alive_bonus = 0.2
    c.com_z_min = 0.5
    c.ctrl_freq = 200.0
    c.ep_dur_max = 3000
    c.mirror_policy = 1
    c.precision = 32
    c.env_index_base = 0
    c.seed = 1234
    c.env_kind = DL_ENV_STRAIGHT
    c.lanes_per_walker = 0          # auto
    c.intended_semantics = 0        # strict_reference_quirks: the reference's behaviour incl. Q1-Q4
    c.strict_solver = 0
    for k, v in kw.items():
        if k == 'rew_weights':
            c.rew_weights[:] = list(v)
        elif k == 'strict_reference_quirks':          # True (default): as the reference; False: every switchable quirk (Q2, Q3, Q4) off
            c.intended_semantics = 0 if v else DL_INTENDED_ALL
        else:
            setattr(c, k, v)
    return c
