#!/usr/bin/env python3
"""mjcf_model.py -- model-constant extractor.

Parses a robot MJCF file (the subset the GUARD robots use: one <default> class, nested bodies,
slide/hinge joints, sphere/capsule/box geoms, motor/general actuators) and derives, in float64 and
without `mujoco`, the quantities MuJoCo's compiler and mj_setConst would hand to MJX:

  * body masses, centres of mass and inertia tensors from geom shapes and densities
  * joint tables: axis, damping, armature, stiffness, range (degrees -> radians), limited
  * M(qpos0) by summing m Jp'Jp + Jr' I Jr over bodies (+ armature), dof_invweight0 = diag(M^-1),
    body_invweight0 = (tr(Jp M^-1 Jp')/3, tr(Jr M^-1 Jr')/3) at each body's centre of mass
  * option timestep, geom margin / friction, contact-capable geoms
  * actuators with MuJoCo's default-class resolution: a class holds ONE actuator default; the
    <general>/<motor>/<position>/<velocity> children of <default> all write it, in document order
    (XML reference, default/motor .. default/velocity: "set the attributes of the general element
    using Actuator shortcuts"); an actuator element starts from that default and applies its own tag
    the same way.  Result per actuator: gear, ctrllimited/ctrlrange, forcelimited/forcerange,
    gaintype/gainprm, biastype/biasprm -- what mjx fwd_actuation reads

[derived]: MuJoCo XML reference + computation chapter (kinematics: the joints of one body are applied
in order, each in the frame left by the previous one).  Used at development time only:

    python tools/model_constants.py /path/to/safe_rl_envs/xmls

prints the constants carried by guardx_amd/csrc/gx_robot*.h (and by the CPU checker).
"""
import xml.etree.ElementTree as ET

import numpy as np

DEG = np.pi / 180.0
ACTUATOR_TAGS = ('general', 'motor', 'position', 'velocity')


def _vec(s, n=None, default=None):
    if s is None:
        return None if default is None else np.asarray(default, float)
    v = np.array([float(x) for x in s.split()], float)
    return v if n is None else v[:n]


def _rot(axis, angle):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


class Model:
    def __init__(self, path):
        root = ET.parse(path).getroot()
        comp = root.find('compiler')
        self.degree = comp is None or comp.get('angle', 'degree') == 'degree'
        opt = root.find('option')
        self.timestep = float(opt.get('timestep', 0.002)) if opt is not None else 0.002
        dflt = root.find('default')
        self.dflt = {k: dict(dflt.find(k).attrib) if dflt is not None and dflt.find(k) is not None else {}
                     for k in ('geom', 'joint', 'motor', 'general')}
        self.bodies = [dict(name='world', parent=-1, pos=np.zeros(3), joints=[], geoms=[])]
        wb = root.find('worldbody')
        for g in wb.findall('geom'):
            self.bodies[0]['geoms'].append(self._geom(g))
        for b in wb.findall('body'):
            self._body(b, 0)
        # ---- actuators: one default per class, shortcuts overwrite it in document order ----------
        act_default = self._new_actuator()
        for el in (list(dflt) if dflt is not None else []):
            if el.tag in ACTUATOR_TAGS:
                self._apply_actuator(act_default, el)
        self.actuator_default = act_default
        self.actuators = []
        act = root.find('actuator')
        for a in (list(act) if act is not None else []):
            d = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in act_default.items()}
            self._apply_actuator(d, a)
            d.update(kind=a.tag, joint=a.get('joint'), name=a.get('name'))
            # autolimits (MuJoCo >= 2.3 default): "auto" means limited iff a range was given
            for lim, rng in (('ctrllimited', 'ctrlrange'), ('forcelimited', 'forcerange')):
                if d[lim] == 'auto':
                    d[lim] = bool(d[rng][0] != 0 or d[rng][1] != 0)
            self.actuators.append(d)
        self._compile()

    @staticmethod
    def _new_actuator():
        """mjs_defaultActuator: fixed gain 1, no bias, gear 1, limits "auto" with empty ranges"""
        return dict(gear=1.0, ctrllimited='auto', forcelimited='auto', ctrlrange=np.zeros(2), forcerange=np.zeros(2),
                    gaintype='fixed', biastype='none', gainprm=np.array([1.0, 0, 0]), biasprm=np.zeros(3))

    @staticmethod
    def _apply_actuator(d, el):
        """mjXReader::OneActuator: common attributes, then the tag's own rule"""
        at = el.attrib
        for lim in ('ctrllimited', 'forcelimited'):
            if lim in at:
                d[lim] = {'true': True, 'false': False, 'auto': 'auto'}[at[lim]]
        for rng in ('ctrlrange', 'forcerange'):
            if rng in at:
                d[rng] = _vec(at[rng], 2)
        if 'gear' in at:
            d['gear'] = _vec(at['gear'])[0]
        if el.tag == 'general':                        # explicit attributes only
            for k in ('gaintype', 'biastype'):
                if k in at:
                    d[k] = at[k]
            for k in ('gainprm', 'biasprm'):
                if k in at:
                    v = _vec(at[k]); d[k] = np.concatenate([v, np.zeros(3)])[:3]
        elif el.tag == 'motor':                        # direct drive: fixed gain 1, no bias
            d.update(gaintype='fixed', biastype='none', biasprm=np.zeros(3))
            d['gainprm'] = d['gainprm'].copy(); d['gainprm'][0] = 1.0
        elif el.tag == 'position':                     # kp (default: the gain already there)
            kp = float(at['kp']) if 'kp' in at else d['gainprm'][0]
            d.update(gaintype='fixed', biastype='affine', biasprm=np.array([0.0, -kp, 0.0]))
            d['gainprm'] = d['gainprm'].copy(); d['gainprm'][0] = kp
        elif el.tag == 'velocity':                     # kv (default: the gain already there)
            kv = float(at['kv']) if 'kv' in at else d['gainprm'][0]
            d.update(gaintype='fixed', biastype='affine', biasprm=np.array([0.0, 0.0, -kv]))
            d['gainprm'] = d['gainprm'].copy(); d['gainprm'][0] = kv
        else:
            raise NotImplementedError("actuator shortcut <%s>" % el.tag)

    # ---- parsing -------------------------------------------------------------------------
    def _attr(self, kind, el):
        at = dict(self.dflt[kind]); at.update(el.attrib)
        return at

    def _geom(self, el):
        at = self._attr('geom', el)
        g = dict(name=at.get('name'), type=at.get('type', 'sphere'), size=_vec(at.get('size'), default=[0.0]),
                 pos=_vec(at.get('pos'), 3, default=[0, 0, 0]), density=float(at.get('density', 1000.0)),
                 fromto=_vec(at.get('fromto'), 6), contype=int(at.get('contype', 1)),
                 conaffinity=int(at.get('conaffinity', 1)), margin=float(at.get('margin', 0.0)),
                 friction=_vec(at.get('friction'), default=[1.0, 0.005, 0.0001]), condim=int(at.get('condim', 3)))
        return g

    def _body(self, el, parent):
        b = dict(name=el.get('name'), parent=parent, pos=_vec(el.get('pos'), 3, default=[0, 0, 0]), joints=[], geoms=[])
        self.bodies.append(b)
        me = len(self.bodies) - 1
        for j in el.findall('joint'):
            at = self._attr('joint', j)
            rng = _vec(at.get('range'), 2)
            limited = at.get('limited', 'auto')
            limited = (rng is not None) if limited == 'auto' else (limited == 'true')
            scale = DEG if (self.degree and at.get('type', 'hinge') == 'hinge') else 1.0
            b['joints'].append(dict(name=at.get('name'), type=at.get('type', 'hinge'),
                                    axis=_vec(at.get('axis'), 3, default=[0, 0, 1]),
                                    pos=_vec(at.get('pos'), 3, default=[0, 0, 0]),
                                    damping=float(at.get('damping', 0.0)), armature=float(at.get('armature', 0.0)),
                                    stiffness=float(at.get('stiffness', 0.0)), limited=limited,
                                    range=None if rng is None else rng * scale))
        for g in el.findall('geom'):
            b['geoms'].append(self._geom(g))
        for c in el.findall('body'):
            self._body(c, me)

    # ---- compile ---------------------------------------------------------------------------
    @staticmethod
    def geom_inertial(g):
        """(mass, com, inertia about the com) in the body frame -- MuJoCo's shape formulas"""
        rho = g['density']
        if g['type'] == 'sphere':
            r = g['size'][0]
            m = rho * 4 / 3 * np.pi * r ** 3
            return m, g['pos'], 0.4 * m * r * r * np.eye(3)
        if g['type'] == 'box':
            hx, hy, hz = g['size'][:3]
            m = rho * 8 * hx * hy * hz
            return m, g['pos'], np.diag([m / 3 * (hy * hy + hz * hz), m / 3 * (hx * hx + hz * hz), m / 3 * (hx * hx + hy * hy)])
        if g['type'] == 'capsule':
            r = g['size'][0]
            if g['fromto'] is not None:
                a, b = g['fromto'][:3], g['fromto'][3:]
                length = np.linalg.norm(b - a); u = (b - a) / length; c = 0.5 * (a + b)
            else:
                length = 2 * g['size'][1]; u = np.array([0, 0, 1.0]); c = g['pos']
            vc, vs = np.pi * r * r * length, 4 / 3 * np.pi * r ** 3
            mc, ms = rho * vc, rho * vs
            it = mc * (3 * r * r + length * length) / 12 + 2 * ms * r * r / 5 + ms * length * (3 * r + 2 * length) / 8
            ia = mc * r * r / 2 + 2 * ms * r * r / 5
            return mc + ms, c, it * np.eye(3) + (ia - it) * np.outer(u, u)
        return 0.0, np.zeros(3), np.zeros((3, 3))          # plane etc.

    def _compile(self):
        nb = len(self.bodies)
        self.mass = np.zeros(nb); self.ipos = np.zeros((nb, 3)); self.inertia = np.zeros((nb, 3, 3))
        for k, b in enumerate(self.bodies):
            parts = [self.geom_inertial(g) for g in b['geoms']] if k else []
            m = sum(p[0] for p in parts)
            if m > 0:
                com = sum(p[0] * p[1] for p in parts) / m
                inertia = sum(pi + pm * (((pc - com) @ (pc - com)) * np.eye(3) - np.outer(pc - com, pc - com))
                              for pm, pc, pi in parts)
                self.mass[k], self.ipos[k], self.inertia[k] = m, com, inertia
        self.dof_body = [k for k, b in enumerate(self.bodies) for _ in b['joints']]
        self.dof_joint = [j for b in self.bodies for j in b['joints']]
        self.nv = len(self.dof_joint)
        q0 = np.zeros(self.nv)
        self.M0 = self.mass_matrix(q0)
        A0 = np.linalg.inv(self.M0)
        self.dof_invweight0 = np.diag(A0).copy()
        self.body_invweight0 = np.zeros((nb, 2))
        kin = self.kinematics(q0)
        for k in range(1, nb):
            if self.mass[k] > 0:
                jp, jr = self.jac(kin, k, kin['xpos'][k] + kin['R'][k] @ self.ipos[k])
                self.body_invweight0[k] = np.trace(jp @ A0 @ jp.T) / 3, np.trace(jr @ A0 @ jr.T) / 3

    def kinematics(self, q):
        nb = len(self.bodies)
        xpos = np.zeros((nb, 3)); R = np.zeros((nb, 3, 3)); R[0] = np.eye(3)
        axis = np.zeros((self.nv, 3)); anchor = np.zeros((self.nv, 3))
        d = 0
        for k in range(1, nb):
            b = self.bodies[k]
            p = xpos[b['parent']] + R[b['parent']] @ b['pos']
            Rk = R[b['parent']].copy()
            for j in b['joints']:
                anchor[d] = p + Rk @ j['pos']
                axis[d] = Rk @ (j['axis'] / np.linalg.norm(j['axis']))
                if j['type'] == 'slide':
                    p = p + axis[d] * q[d]
                else:
                    Rk = _rot(axis[d], q[d]) @ Rk
                    p = anchor[d] - Rk @ j['pos']
                d += 1
            xpos[k], R[k] = p, Rk
        return dict(xpos=xpos, R=R, axis=axis, anchor=anchor)

    def jac(self, kin, body, point):
        jp = np.zeros((3, self.nv)); jr = np.zeros((3, self.nv))
        chain = set()
        b = body
        while b > 0:
            chain.add(b); b = self.bodies[b]['parent']
        for d in range(self.nv):
            if self.dof_body[d] not in chain:
                continue
            if self.dof_joint[d]['type'] == 'slide':
                jp[:, d] = kin['axis'][d]
            else:
                jp[:, d] = np.cross(kin['axis'][d], point - kin['anchor'][d]); jr[:, d] = kin['axis'][d]
        return jp, jr

    def mass_matrix(self, q):
        kin = self.kinematics(q)
        M = np.diag([j['armature'] for j in self.dof_joint]).astype(float)
        for k in range(1, len(self.bodies)):
            if self.mass[k] > 0:
                jp, jr = self.jac(kin, k, kin['xpos'][k] + kin['R'][k] @ self.ipos[k])
                Iw = kin['R'][k] @ self.inertia[k] @ kin['R'][k].T
                M = M + self.mass[k] * jp.T @ jp + jr.T @ Iw @ jr
        return M

    def body(self, name):
        return next(k for k, b in enumerate(self.bodies) if b['name'] == name)

    def summary(self):
        out = ["timestep %.6g   nbody %d   nv %d   angle=%s" % (self.timestep, len(self.bodies), self.nv,
                                                              "degree" if self.degree else "radian")]
        for k, b in enumerate(self.bodies):
            if k == 0:
                continue
            out.append("body %-18s parent %-16s pos %s mass %.17g" % (b['name'], self.bodies[b['parent']]['name'],
                                                                   b['pos'].tolist(), self.mass[k]))
            for j in b['joints']:
                out.append("    joint %-12s %-5s axis %s damping %g armature %g stiffness %g limited %s range %s" % (
                    j['name'], j['type'], j['axis'].tolist(), j['damping'], j['armature'], j['stiffness'], j['limited'],
                    None if j['range'] is None else ["%.17g" % x for x in j['range']]))
        out.append("dof_invweight0 " + " ".join("%.17g" % x for x in self.dof_invweight0))
        for a in self.actuators:
            out.append("actuator %-8s joint %-12s gear %g ctrllimited %s ctrlrange %s forcelimited %s forcerange %s "
                       "gain %s %s bias %s %s" % (
                           a['kind'], a['joint'], a['gear'], a['ctrllimited'], a['ctrlrange'].tolist(), a['forcelimited'],
                           a['forcerange'].tolist(), a['gaintype'], a['gainprm'].tolist(), a['biastype'],
                           a['biasprm'].tolist()))
        return "\n".join(out)


if __name__ == "__main__":
    import sys
    print(Model(sys.argv[1]).summary())
