"""MJCF subset -> dl_model_desc (host side, init time only).

Covers exactly what drloco/mujoco/xml/walker3d_flat_feet.xml and walker_165cm_65kg.xml use:
`compiler angle=radian coordinate=local inertiafromgeom=false`, `default` classes for
joint/motor/geom, `option integrator=RK4 timestep`, a floor plane at z = 0, nested bodies with
explicit `inertial` (diaginertia), slide/hinge joints, capsule (fromto) / box (pos, axisangle)
geoms, sites, and `motor` actuators.  Everything else raises.

`finalize()` computes the constants MuJoCo derives at compile time (mj_setConst at qpos0):
dof_invweight0, body_invweight0 and stat.meaninertia, with a small numpy forward-kinematics /
mass-matrix evaluation.
"""
import xml.etree.ElementTree as ET

import numpy as np

from . import abi


def _f(s, n=None, default=None):
    if s is None:
        return default
    v = [float(x) for x in s.split()]
    if n is not None and len(v) != n:
        raise ValueError(f'expected {n} numbers, got {s!r}')
    return v


def _axisangle_mat(axis, ang):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s, t = np.cos(ang), np.sin(ang), 1 - np.cos(ang)
    return np.array([[t * x * x + c, t * x * y - s * z, t * x * z + s * y],
                     [t * x * y + s * z, t * y * y + c, t * y * z - s * x],
                     [t * x * z - s * y, t * y * z + s * x, t * z * z + c]])


def _z_to_vec_mat(vec):
    """Rotation taking the z axis to `vec` (what the MuJoCo compiler does for fromto geoms)."""
    v = np.asarray(vec, float)
    v = v / np.linalg.norm(v)
    z = np.array([0.0, 0.0, 1.0])
    ax = np.cross(z, v)
    s = np.linalg.norm(ax)
    if s < 1e-10:
        return np.eye(3) if v[2] > 0 else _axisangle_mat([1, 0, 0], np.pi)
    return _axisangle_mat(ax / s, np.arctan2(s, v[2]))


class ModelBuilder:
    """Accumulates bodies/joints/geoms/sites/motors and emits an abi.ModelDesc."""

    def __init__(self, timestep=0.001, frame_skip=5):
        self.timestep = timestep
        self.frame_skip = frame_skip
        self.bodies = [dict(name='world', parent=0, pos=[0, 0, 0], mass=0.0, ipos=[0, 0, 0], inertia=[0, 0, 0])]
        self.joints, self.geoms, self.sites, self.motors = [], [], [], []
        self.floor_friction = 0.7

    def body(self, name, parent, pos, mass, ipos, inertia):
        self.bodies.append(dict(name=name, parent=parent, pos=list(pos), mass=mass, ipos=list(ipos), inertia=list(inertia)))
        return len(self.bodies) - 1

    def joint(self, name, body, jtype, axis, pos=(0, 0, 0), ref=0.0, limited=False, range=(0, 0), damping=0.0, armature=0.0):
        self.joints.append(dict(name=name, body=body, type=jtype, axis=list(axis), pos=list(pos), ref=ref,
                                limited=bool(limited), range=list(range), damping=damping, armature=armature))
        return len(self.joints) - 1

    def capsule(self, body, fromto, radius, friction):
        a, b = np.array(fromto[:3], float), np.array(fromto[3:], float)
        self.geoms.append(dict(type=abi.DL_GEOM_CAPSULE, body=body, pos=list((a + b) / 2), mat=_z_to_vec_mat(b - a),
                               size=[radius, np.linalg.norm(b - a) / 2, 0.0], friction=friction))

    def box(self, body, pos, half, friction, axisangle=None):
        mat = np.eye(3) if axisangle is None else _axisangle_mat(axisangle[:3], axisangle[3])
        self.geoms.append(dict(type=abi.DL_GEOM_BOX, body=body, pos=list(pos), mat=mat, size=list(half), friction=friction))

    def site(self, body, pos):
        self.sites.append(dict(body=body, pos=list(pos)))

    def motor(self, joint_name, gear=1.0, ctrlrange=(-300, 300), forcerange=(-300, 300)):
        idx = [j['name'] for j in self.joints].index(joint_name)
        self.motors.append(dict(dof=idx, gear=gear, ctrlrange=list(ctrlrange), forcerange=list(forcerange)))

    def build(self):
        m = abi.ModelDesc()
        if len(self.bodies) > abi.DL_MAX_BODY or len(self.joints) > abi.DL_MAX_DOF or len(self.geoms) > abi.DL_MAX_GEOM \
                or len(self.sites) > abi.DL_MAX_SITE or len(self.motors) > abi.DL_MAX_ACT:
            raise ValueError('model exceeds the static capacities of dl_model_desc')
        m.nbody, m.nv, m.nu, m.ngeom, m.nsite = len(self.bodies), len(self.joints), len(self.motors), len(self.geoms), len(self.sites)
        m.frame_skip, m.timestep = self.frame_skip, self.timestep
        m.gravity[:] = [0, 0, -9.81]
        m.solref[:] = [0.02, 1.0]
        m.solimp[:] = [0.9, 0.95, 0.001, 0.5, 2.0]
        m.tolerance, m.ls_tolerance, m.iterations, m.ls_iterations = 1e-8, 0.01, 100, 50
        for b, bd in enumerate(self.bodies):
            m.body_parent[b] = bd['parent']
            m.body_pos[b][:] = bd['pos']
            m.body_mass[b] = bd['mass']
            m.body_ipos[b][:] = bd['ipos']
            m.body_inertia[b][:] = bd['inertia']
        last_body = 0
        for j, jd in enumerate(self.joints):
            if jd['body'] < last_body:
                raise ValueError("a body's joints must be contiguous and in body order")
            last_body = jd['body']
            m.jnt_type[j], m.jnt_body[j] = jd['type'], jd['body']
            ax = np.asarray(jd['axis'], float)
            m.jnt_axis[j][:] = list(ax / np.linalg.norm(ax))
            m.jnt_pos[j][:] = jd['pos']
            m.jnt_qpos0[j] = jd['ref']
            m.jnt_limited[j] = int(jd['limited'])
            m.jnt_range[j][:] = jd['range']
            m.jnt_damping[j], m.jnt_armature[j] = jd['damping'], jd['armature']
        for g, gd in enumerate(self.geoms):
            m.geom_type[g], m.geom_body[g] = gd['type'], gd['body']
            m.geom_pos[g][:] = gd['pos']
            m.geom_mat[g][:] = list(np.asarray(gd['mat'], float).reshape(9))
            m.geom_size[g][:] = gd['size']
            m.geom_friction[g] = gd['friction']
        m.floor_friction = self.floor_friction
        for s, sd in enumerate(self.sites):
            m.site_body[s] = sd['body']
            m.site_pos[s][:] = sd['pos']
        for a, ad in enumerate(self.motors):
            m.act_dof[a], m.act_gear[a] = ad['dof'], ad['gear']
            m.act_ctrlrange[a][:] = ad['ctrlrange']
            m.act_forcerange[a][:] = ad['forcerange']
        finalize(m)
        return m


# ---------------------------------------------------------------------------------------------
def _kinematics(m, q):
    nb, nv = m.nbody, m.nv
    xpos, xmat = np.zeros((nb, 3)), np.tile(np.eye(3), (nb, 1, 1))
    anchor, axis = np.zeros((nv, 3)), np.zeros((nv, 3))
    for b in range(1, nb):
        p = m.body_parent[b]
        pos = xpos[p] + xmat[p] @ np.array(m.body_pos[b][:])
        R = xmat[p].copy()
        for j in range(nv):
            if m.jnt_body[j] != b:
                continue
            jp, ja = np.array(m.jnt_pos[j][:]), np.array(m.jnt_axis[j][:])
            anchor[j], axis[j] = pos + R @ jp, R @ ja
            dq = q[j] - m.jnt_qpos0[j]
            if m.jnt_type[j] == abi.DL_JNT_SLIDE:
                pos = pos + axis[j] * dq
            else:
                R = R @ _axisangle_mat(ja, dq)
                pos = anchor[j] - R @ jp
        xpos[b], xmat[b] = pos, R
    return xpos, xmat, anchor, axis


def _ancestors(m):
    anc = np.zeros((m.nbody, m.nv), bool)
    for b in range(1, m.nbody):
        a = b
        while a > 0:
            for j in range(m.nv):
                if m.jnt_body[j] == a:
                    anc[b, j] = True
            a = m.body_parent[a]
    return anc


def _jac(m, anc, anchor, axis, point, b):
    jp, jr = np.zeros((3, m.nv)), np.zeros((3, m.nv))
    for j in range(m.nv):
        if not anc[b, j]:
            continue
        if m.jnt_type[j] == abi.DL_JNT_SLIDE:
            jp[:, j] = axis[j]
        else:
            jp[:, j] = np.cross(axis[j], point - anchor[j])
            jr[:, j] = axis[j]
    return jp, jr


def mass_matrix(m, q):
    """Joint-space inertia at configuration q (numpy, init-time use only)."""
    xpos, xmat, anchor, axis = _kinematics(m, q)
    anc = _ancestors(m)
    M = np.zeros((m.nv, m.nv))
    jacs = {}
    for b in range(1, m.nbody):
        com = xpos[b] + xmat[b] @ np.array(m.body_ipos[b][:])
        jp, jr = _jac(m, anc, anchor, axis, com, b)
        Iw = xmat[b] @ np.diag(m.body_inertia[b][:]) @ xmat[b].T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
        jacs[b] = (jp, jr)
    M += np.diag(m.jnt_armature[:m.nv])
    return M, jacs


def finalize(m):
    """mj_setConst: dof_invweight0, body_invweight0, meaninertia at qpos0."""
    q0 = np.array(m.jnt_qpos0[:m.nv])
    M, jacs = mass_matrix(m, q0)
    Minv = np.linalg.inv(M)
    for j in range(m.nv):
        m.dof_invweight0[j] = Minv[j, j]
    m.meaninertia = float(np.trace(M) / m.nv)
    m.body_invweight0[0][:] = [0.0, 0.0]
    for b in range(1, m.nbody):
        jp, jr = jacs[b]
        m.body_invweight0[b][0] = float(np.trace(jp @ Minv @ jp.T) / 3)
        m.body_invweight0[b][1] = float(np.trace(jr @ Minv @ jr.T) / 3)


# ---------------------------------------------------------------------------------------------
def parse_mjcf(path, frame_skip):
    """Parse an MJCF file of the supported subset into an abi.ModelDesc.

    frame_skip = sim_freq / CTRL_FREQ (drloco/mujoco/mimic_env.py:194-207)."""
    root = ET.parse(path).getroot()
    comp = root.find('compiler')
    if comp is None or comp.get('angle') != 'radian' or comp.get('coordinate', 'local') != 'local' \
            or comp.get('inertiafromgeom', 'auto') != 'false':
        raise ValueError('unsupported <compiler> settings')
    opt = root.find('option')
    if opt is None or opt.get('integrator') != 'RK4':
        raise ValueError('only integrator="RK4" is supported')
    dflt = root.find('default')
    dj = dict(dflt.find('joint').attrib) if dflt is not None and dflt.find('joint') is not None else {}
    dm = dict(dflt.find('motor').attrib) if dflt is not None and dflt.find('motor') is not None else {}
    dg = dict(dflt.find('geom').attrib) if dflt is not None and dflt.find('geom') is not None else {}
    mb = ModelBuilder(timestep=float(opt.get('timestep', 0.002)), frame_skip=frame_skip)

    def attr(el, d, k, default=None):
        return el.get(k, d.get(k, default))

    wb = root.find('worldbody')
    floors = [g for g in wb.findall('geom') if g.get('type') == 'plane']
    if len(floors) != 1 or _f(floors[0].get('pos', '0 0 0')) != [0, 0, 0] or int(attr(floors[0], dg, 'conaffinity', 1)) != 1:
        raise ValueError('exactly one floor plane at the origin is supported')
    mb.floor_friction = _f(attr(floors[0], dg, 'friction'))[0]

    def walk(el, parent):
        inert = el.find('inertial')
        if inert is None or inert.get('diaginertia') is None:
            raise ValueError('bodies need an explicit <inertial diaginertia=...>')
        b = mb.body(el.get('name'), parent, _f(el.get('pos', '0 0 0'), 3), float(inert.get('mass')),
                    _f(inert.get('pos', '0 0 0'), 3), _f(inert.get('diaginertia'), 3))
        if el.get('quat') or el.get('axisangle') or el.get('euler'):
            raise ValueError('rotated body frames are not supported')
        for j in el.findall('joint'):
            jt = attr(j, dj, 'type', 'hinge')
            if jt not in ('slide', 'hinge'):
                raise ValueError(f'unsupported joint type {jt}')
            if float(attr(j, dj, 'stiffness', 0)) != 0 or float(attr(j, dj, 'frictionloss', 0)) != 0:
                raise ValueError('joint stiffness/frictionloss are not supported')
            mb.joint(j.get('name'), b, abi.DL_JNT_SLIDE if jt == 'slide' else abi.DL_JNT_HINGE,
                     _f(attr(j, dj, 'axis', '0 0 1'), 3), _f(attr(j, dj, 'pos', '0 0 0'), 3),
                     float(attr(j, dj, 'ref', 0)), attr(j, dj, 'limited', 'false') == 'true',
                     _f(attr(j, dj, 'range', '0 0'), 2), float(attr(j, dj, 'damping', 0)),
                     float(attr(j, dj, 'armature', 0)))
        for g in el.findall('geom'):
            if int(attr(g, dg, 'contype', 1)) != 1 or int(attr(g, dg, 'conaffinity', 1)) != 0 or int(attr(g, dg, 'condim', 3)) != 3:
                raise ValueError('body geoms must have contype=1 conaffinity=0 condim=3')
            fr = _f(attr(g, dg, 'friction'))[0]
            gt = attr(g, dg, 'type', 'sphere')
            size = _f(g.get('size'))
            if gt == 'capsule':
                mb.capsule(b, _f(g.get('fromto'), 6), size[0], fr)
            elif gt == 'box':
                aa = _f(g.get('axisangle'), 4) if g.get('axisangle') else None
                mb.box(b, _f(g.get('pos', '0 0 0'), 3), size, fr, aa)
            else:
                raise ValueError(f'unsupported geom type {gt}')
        for s in el.findall('site'):
            mb.site(b, _f(s.get('pos', '0 0 0'), 3))
        for c in el.findall('body'):
            walk(c, b)

    for el in wb.findall('body'):
        walk(el, 0)
    for a in root.find('actuator'):
        if a.tag != 'motor':
            raise ValueError('only <motor> actuators are supported')
        cr = _f(attr(a, dm, 'ctrlrange', '0 0'), 2) if attr(a, dm, 'ctrllimited', 'false') == 'true' else [-1e30, 1e30]
        frg = _f(attr(a, dm, 'forcerange', '0 0'), 2) if attr(a, dm, 'forcelimited', 'false') == 'true' else [-1e30, 1e30]
        mb.motor(a.get('joint'), float(attr(a, dm, 'gear', '1').split()[0]), cr, frg)
    return mb.build()
