"""Built-in walker models (the reference's `env_map`, drloco/mujoco/config.py:9-14).

The straight walker's constants are stated here through ModelBuilder calls (values from
drloco/mujoco/xml/walker3d_flat_feet.xml, listed in SURVEY.md appendix D) so that the package
runs on a machine that has no copy of the reference; `mjcf.parse_mjcf` reads the same model
from an MJCF file when one is supplied (tests compare both).
"""
from . import abi
from .mjcf import ModelBuilder

STRAIGHT_WALKER = 'StraightMimicWalker'       # drloco/mujoco/config.py:5
WALKER_165CM = 'MimicWalker165cm65kg'         # drloco/mujoco/config.py:6

SIM_FREQ = {STRAIGHT_WALKER: 1000, WALKER_165CM: 1000}    # drloco/mujoco/config.py:13-14
CTRL_FREQ = {STRAIGHT_WALKER: 200, WALKER_165CM: 100}      # drloco/config/config.py:20-21
ENV_KIND = {STRAIGHT_WALKER: abi.DL_ENV_STRAIGHT, WALKER_165CM: abi.DL_ENV_LOCO3D}
GAMMA = {STRAIGHT_WALKER: 0.995, WALKER_165CM: 0.99}       # drloco/config/hypers.py:68

SLIDE, HINGE = abi.DL_JNT_SLIDE, abi.DL_JNT_HINGE


def _leg(mb, torso, side):
    """side = -1 right, +1 left (walker3d_flat_feet.xml:26-68)."""
    r = side < 0
    hip_front = (-0.7854, 0.0873) if r else (-0.0873, 0.7854)
    names = (('right_thigh', 'right_shank', 'right_foot', 'hip_joint_saggital_right', 'hip_joint_frontal_right',
              'knee_joint_right', 'ankle_joint_right') if r else
             ('thigh_left', 'shank_left', 'foot_left', 'hip_joint_saggital_left', 'hip_joint_frontal_left',
              'knee_left_joint', 'ankle_left_joint'))
    thigh = mb.body(names[0], torso, (0, 0.08 * side, 0), 8.5, (0, 0, -0.2), (0.15, 0.15, 0.03))
    mb.joint(names[3], thigh, HINGE, (0, 1, 0), limited=True, range=(-0.8727, 0.8727), damping=28, armature=0.01)
    mb.joint(names[4], thigh, HINGE, (1, 0, 0), limited=True, range=hip_front, damping=28, armature=0.01)
    mb.capsule(thigh, (0, 0, -0.05, 0, 0, -0.45), 0.05, 0.9)
    shank = mb.body(names[1], thigh, (0, 0, -0.5), 3.5, (0, 0, -0.2), (0.05, 0.05, 0.003))
    mb.joint(names[5], shank, HINGE, (0, 1, 0), limited=True, range=(0.0, 2.6180), damping=12, armature=0.01)
    mb.capsule(shank, (0, 0, -0.05, 0, 0, -0.45), 0.04, 0.9)
    foot = mb.body(names[2], shank, (0, 0, -0.5), 1.5, (0.06, 0, -0.07), (0.003, 0.006, 0.005))
    mb.joint(names[6], foot, HINGE, (0, 1, 0), limited=True, range=(-0.3491, 0.6981), damping=20, armature=0.01)
    mb.box(foot, (0.0675, 0.005 * side, -0.04), (0.11, 0.05, 0.04), 0.9, axisangle=(0, 0, 1, 0.05 * side))
    # foot-sole corner sites: front-left, front-right, back-left, back-right
    fl, fr = (0.04, -0.06) if r else (0.06, -0.04)
    for x, y in ((0.1775, fl), (0.1775, fr), (-0.0425, 0.05), (-0.0425, -0.05)):
        mb.site(foot, (x, y, -0.08))
    return names[3:]


def walker3d_flat_feet():
    """nq = nv = 14, nu = 8, 80.5 kg; qpos order: com x,y,z, trunk rx,ry,rz, right hip sag/front,
    knee, ankle, left hip sag/front, knee, ankle."""
    mb = ModelBuilder(timestep=0.001, frame_skip=SIM_FREQ[STRAIGHT_WALKER] // CTRL_FREQ[STRAIGHT_WALKER])
    mb.floor_friction = 0.7
    torso = mb.body('torso', 0, (0, 0, 1.08), 53.5, (0, 0, 0.35), (2.5, 4.0, 1.5))
    mb.joint('com_x', torso, SLIDE, (1, 0, 0))
    mb.joint('com_y', torso, SLIDE, (0, 1, 0))
    mb.joint('com_z', torso, SLIDE, (0, 0, 1), pos=(0, 0, -1.08), ref=1.08)
    mb.joint('trunk_rot_x', torso, HINGE, (1, 0, 0))
    mb.joint('trunk_rot_y', torso, HINGE, (0, 1, 0))
    mb.joint('trunk_rot_z', torso, HINGE, (0, 0, 1))
    mb.capsule(torso, (0, 0, 0, 0, 0, 0.7), 0.075, 0.9)
    motors = _leg(mb, torso, -1) + _leg(mb, torso, +1)
    for name in motors:
        mb.motor(name, gear=1.0, ctrlrange=(-300, 300), forcerange=(-300, 300))
    return mb.build()


def _leg165(mb, pelvis, side):
    """walker_165cm_65kg.xml:34-78; side = -1 right, +1 left."""
    r = side < 0
    sfx = 'r' if r else 'l'
    thigh = mb.body('thigh_right' if r else 'thigh_left', pelvis, (0, 0.08 * side, 0), 6.9, (0, 0, -0.2136), (0.122, 0.122, 0.024))
    mb.joint(f'hip_flexion_{sfx}', thigh, HINGE, (0, -1, 0), limited=True, range=(-0.8727, 0.8727), damping=28, armature=0.01)
    mb.joint(f'hip_adduction_{sfx}', thigh, HINGE, (1, 0, 0) if r else (-1, 0, 0), limited=True, range=(-0.7854, 0.0873), damping=28, armature=0.01)
    mb.joint(f'hip_rotation_{sfx}', thigh, HINGE, (0, 0, -1), limited=True, range=(-0.26, 0.26), damping=28, armature=0.01)
    mb.capsule(thigh, (0, 0, -0.05, 0, 0, -0.4272), 0.05, 0.9)
    shank = mb.body('shank_right' if r else 'shank_left', thigh, (0, 0, -0.4772), 2.8, (0, 0, -0.2136), (0.04, 0.04, 0.0024))
    mb.joint(f'knee_angle_{sfx}', shank, HINGE, (0, -1, 0), limited=True, range=(-2.6180, 0.0), damping=12, armature=0.01)
    mb.capsule(shank, (0, 0, -0.05, 0, 0, -0.4272 if r else -0.45), 0.04, 0.9)
    foot = mb.body('foot_right' if r else 'foot_left', shank, (0, 0, -0.4772), 1.2, (0.06, 0, -0.07), (0.003, 0.006, 0.005))
    if r:
        mb.joint('ankle_angle_r', foot, HINGE, (0, 1, 0), limited=True, range=(-0.3491, 0.6981), damping=20, armature=0.01)
    else:
        mb.joint('ankle_angle_l', foot, HINGE, (0, -1, 0), limited=True, range=(-0.6981, 0.3491), damping=20, armature=0.01)
    mb.box(foot, (0.0675, 0.005 * side, -0.04), (0.11, 0.05, 0.04), 0.9, axisangle=(0, 0, 1, 0.05 * side))
    fl, fr = (0.04, -0.06) if r else (0.06, -0.04)
    for x, y in ((0.1775, fl), (0.1775, fr), (-0.0425, 0.05), (-0.0425, -0.05)):
        mb.site(foot, (x, y, -0.08))
    return [f'hip_flexion_{sfx}', f'hip_adduction_{sfx}', f'hip_rotation_{sfx}', f'knee_angle_{sfx}', f'ankle_angle_{sfx}']


def walker_165cm_65kg():
    """nq = nv = 19, nu = 13, 65.17 kg (walker_165cm_65kg.xml): pelvis (3 slides + 3 hinges, partly
    negated axes), torso on three lumbar hinges, two legs with 3-dof hips; boxes on pelvis and torso."""
    mb = ModelBuilder(timestep=0.001, frame_skip=SIM_FREQ[WALKER_165CM] // CTRL_FREQ[WALKER_165CM])
    mb.floor_friction = 0.7
    pelvis = mb.body('pelvis', 0, (0, 0, 1.035), 10.87, (0, 0, 0), (0.51, 0.82, 0.31))
    mb.box(pelvis, (0, 0, 0.05), (0.05, 0.1, 0.04), 0.9)
    mb.joint('pelvis_tx', pelvis, SLIDE, (1, 0, 0))
    mb.joint('pelvis_tz', pelvis, SLIDE, (0, -1, 0))
    mb.joint('pelvis_ty', pelvis, SLIDE, (0, 0, 1), ref=1.035)
    mb.joint('pelvis_list', pelvis, HINGE, (1, 0, 0))
    mb.joint('pelvis_tilt', pelvis, HINGE, (0, -1, 0))
    mb.joint('pelvis_rotation', pelvis, HINGE, (0, 0, 1))
    torso = mb.body('torso', pelvis, (0, 0, 0.1075), 32.5, (0, 0, 0.2475), (1.875, 3.0, 1.125))
    mb.box(torso, (0, 0, 0.2475), (0.05, 0.12, 0.2475), 0.9)
    mb.joint('lumbar_bending', torso, HINGE, (1, 0, 0), limited=True, range=(-0.2, 0.15))
    mb.joint('lumbar_extension', torso, HINGE, (0, -1, 0), limited=True, range=(-0.15, 0.15))
    mb.joint('lumbar_rotation', torso, HINGE, (0, 0, 1), limited=True, range=(-0.15, 0.15))
    legs = _leg165(mb, pelvis, -1) + _leg165(mb, pelvis, +1)
    for name in ['lumbar_extension', 'lumbar_bending', 'lumbar_rotation'] + legs:
        mb.motor(name, gear=1.0, ctrlrange=(-300, 300), forcerange=(-300, 300))
    return mb.build()


MODEL_BUILDERS = {STRAIGHT_WALKER: walker3d_flat_feet, WALKER_165CM: walker_165cm_65kg}


def make_model(env_id=STRAIGHT_WALKER):
    try:
        return MODEL_BUILDERS[env_id]()
    except KeyError:
        raise ValueError(f'unknown env id {env_id!r}; available: {sorted(MODEL_BUILDERS)}') from None
