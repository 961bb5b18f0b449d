"""ant_np.py -- TEST INFRASTRUCTURE ONLY (float64, generic).

Independent numpy restatement of one `mjx.step` for xmls/ant.xml ("ant_tiny": planar base
x-slide / z-hinge / y-slide + 4 legs x (hip z-hinge, ankle hinge), 4 foot spheres against the
floor plane).  It is written the GENERIC way on purpose -- body/joint/geom tables, MuJoCo's
kinematics rule, point Jacobians, velocity-product terms by differentiating the Jacobians
numerically, a dense constraint solve -- so that it shares nothing with the closed-form fp32
code in oracle/gx_oracle.c and guardx_amd/csrc/gx_robot.h, which it cross-checks.
It also produces the model constants those two carry (tools/model_constants.py prints them).

PARITY UNPINNED / [derived]: mujoco + mujoco.mjx are absent here, so every rule below follows
the published MuJoCo computation chapter and the MJX sources as documented in DESIGN.md:
  * compile: geom mass/inertia (sphere, capsule), body COM/inertia, degrees -> radians,
    mj_setConst (dof_invweight0 = diag M(qpos0)^-1, body_invweight0 = tr(J M^-1 J^T)/3 at the COM)
  * kinematics: joints of one body are applied in order, each in the frame left by the previous
  * forward: M (incl. armature), bias (velocity products and gravity), passive (damping, spring),
    motors gear*clip(ctrl)
  * constraints: joint-limit rows and sphere-plane contacts with pyramidal friction cones,
    impedance/reference from solref/solimp (refsafe), R = (1-imp)/imp * diagApprox
  * the convex constraint problem is solved to convergence; Euler with implicit joint damping.
Reference lines: xmls/ant.xml (whole file), engine.py:659-700 (one mjx.step per control step).
"""
import numpy as np

DEG = np.pi / 180.0

# ---------------------------------------------------------------------------------------
# model tables restated from xmls/ant.xml
# ---------------------------------------------------------------------------------------
H = 0.09                    # ant.xml:2
RHO = 5.0                   # :6 density
FRICTION = 0.75             # :6 (both geoms; max rule)
MARGIN = 0.01               # :6 (both geoms; max rule)
GEAR = 70.0                 # :7
GRAVITY = np.array([0.0, 0.0, -9.81])   # MuJoCo default (no <option gravity> in the file)
SOLREF = (0.02, 1.0)        # MuJoCo default
SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
LEG_SIGN = [(1, 1), (-1, 1), (-1, -1), (1, -1)]         # :21,38,55,71 fromto signs
ANKLE_AXIS = [(-1, 1, 0), (1, 1, 0), (-1, 1, 0), (1, 1, 0)]   # :27,44,61,77
ANKLE_RANGE = [(30, 70), (-70, -30), (-70, -30), (30, 70)]    # same lines
HIP_RANGE = (-30, 30)


def _unit(v):
    v = np.asarray(v, float)
    return v / np.linalg.norm(v)


class Body:
    def __init__(self, name, parent, pos):
        self.name, self.parent, self.pos = name, parent, np.asarray(pos, float)
        self.joints, self.geoms = [], []


def build_tables():
    """bodies in MJCF depth-first order; joint = dict(type, axis, pos, damping, armature, stiffness, range)"""
    bodies = [Body('world', -1, (0, 0, 0))]
    robot = Body('robot', 0, (0, 0, 0.15))
    robot.joints = [dict(type='slide', axis=(1, 0, 0), damping=0.1, armature=0.0, stiffness=0.0, range=None),
                    dict(type='hinge', axis=(0, 0, 1), damping=0.01, armature=0.0, stiffness=0.1, range=None),
                    dict(type='slide', axis=(0, 1, 0), damping=0.1, armature=0.0, stiffness=0.0, range=None)]
    robot.geoms = [dict(type='sphere', pos=(0, 0, 0), r=0.06, contact=False)]
    bodies.append(robot)
    for i, (sx, sy) in enumerate(LEG_SIGN):
        d05 = np.array([0.05 * sx, 0.05 * sy, 0.0])
        leg = Body('leg%d' % (i + 1), 1, (0, 0, 0))
        leg.geoms = [dict(type='capsule', a=(0, 0, 0), b=d05, r=0.02, contact=False)]
        bodies.append(leg)
        il = len(bodies) - 1
        aux = Body('aux_%d' % (i + 1), il, d05)
        aux.joints = [dict(type='hinge', axis=(0, 0, 1), damping=1.0, armature=1.0, stiffness=0.0,
                           range=(HIP_RANGE[0] * DEG, HIP_RANGE[1] * DEG))]
        aux.geoms = [dict(type='capsule', a=(0, 0, 0), b=d05, r=0.02, contact=False)]
        bodies.append(aux)
        ia = len(bodies) - 1
        ank = Body('ankle_%d' % (i + 1), ia, d05)
        ank.joints = [dict(type='hinge', axis=_unit(ANKLE_AXIS[i]), damping=1.0, armature=1.0, stiffness=0.0,
                           range=(ANKLE_RANGE[i][0] * DEG, ANKLE_RANGE[i][1] * DEG))]
        ank.geoms = [dict(type='capsule', a=(0, 0, 0), b=2 * d05, r=0.02, contact=False),
                     dict(type='sphere', pos=2 * d05, r=0.02, contact=True)]
        bodies.append(ank)
    return bodies


def _geom_inertial(g, RHO=RHO):
    """(mass, com, inertia tensor about the com) in the body frame; MuJoCo geom formulas"""
    if g['type'] == 'sphere':
        m = RHO * 4 / 3 * np.pi * g['r'] ** 3
        return m, np.asarray(g['pos'], float), 0.4 * m * g['r'] ** 2 * np.eye(3)
    a, b, r = np.asarray(g['a'], float), np.asarray(g['b'], float), g['r']
    length = np.linalg.norm(b - a)
    u = (b - a) / length
    vc, vs = np.pi * r * r * length, 4 / 3 * np.pi * r ** 3
    mc, ms = RHO * vc, RHO * vs
    it = mc * (3 * r * r + length * length) / 12 + 2 * ms * r * r / 5 + ms * length * (3 * r + 2 * length) / 8
    ia = mc * r * r / 2 + 2 * ms * r * r / 5
    return mc + ms, 0.5 * (a + b), it * np.eye(3) + (ia - it) * np.outer(u, u)


class TreeModel:
    """generic float64 rigid-body tree (slide/hinge joints, sphere/capsule geoms, foot spheres on a floor plane)"""

    def __init__(self, bodies, h, rho, friction, margin, gear):
        self.bodies = bodies
        self.h, self.friction, self.margin = h, friction, margin
        nb = len(self.bodies)
        self.mass = np.zeros(nb)
        self.ipos = np.zeros((nb, 3))
        self.inertia = np.zeros((nb, 3, 3))
        for k, b in enumerate(self.bodies):
            parts = [_geom_inertial(g, rho) for g in b.geoms]
            m = sum(p[0] for p in parts)
            if m == 0:
                continue
            com = sum(p[0] * p[1] for p in parts) / m
            inertia = np.zeros((3, 3))
            for pm, pc, pi_ in parts:
                d = pc - com
                inertia += pi_ + pm * ((d @ d) * np.eye(3) - np.outer(d, d))
            self.mass[k], self.ipos[k], self.inertia[k] = m, com, inertia
        # dof tables
        self.dof_body, self.dof_joint = [], []
        for k, b in enumerate(self.bodies):
            for j in b.joints:
                self.dof_body.append(k)
                self.dof_joint.append(j)
        self.nv = len(self.dof_body)
        self.damping = np.array([j['damping'] for j in self.dof_joint])
        self.armature = np.array([j['armature'] for j in self.dof_joint])
        self.stiffness = np.array([j['stiffness'] for j in self.dof_joint])
        self.gear = np.zeros(self.nv)
        self.gear[self.nv - len(gear):] = gear              # the motors drive the trailing (leg) DOFs in order
        self.foot = [(k, np.asarray(g['pos'], float), g['r']) for k, b in enumerate(self.bodies)
                     for g in b.geoms if g.get('contact')]
        # mj_setConst at qpos0 = 0
        q0 = np.zeros(self.nv)
        M0 = self.mass_matrix(q0)
        A0 = np.linalg.inv(M0)
        self.dof_invweight0 = np.diag(A0).copy()
        self.body_invweight0 = np.zeros((nb, 2))
        kin = self.kinematics(q0)
        for k in range(1, nb):
            if self.mass[k] == 0:
                continue
            jp, jr = self.jac(kin, k, kin['xpos'][k] + kin['R'][k] @ self.ipos[k])
            self.body_invweight0[k, 0] = np.trace(jp @ A0 @ jp.T) / 3
            self.body_invweight0[k, 1] = np.trace(jr @ A0 @ jr.T) / 3

    # -- kinematics: MuJoCo rule, joints of a body applied in order -----------------------
    def kinematics(self, q):
        nb = len(self.bodies)
        xpos = np.zeros((nb, 3)); R = np.zeros((nb, 3, 3)); R[0] = np.eye(3)
        axis = np.zeros((self.nv, 3)); anchor = np.zeros((self.nv, 3))
        d = 0
        for k in range(1, nb):
            b = self.bodies[k]
            p = xpos[b.parent] + R[b.parent] @ b.pos
            Rk = R[b.parent].copy()
            for j in b.joints:
                jpos = np.asarray(j.get('pos', (0.0, 0.0, 0.0)), float)
                anchor[d] = p + Rk @ jpos
                axis[d] = Rk @ np.asarray(j['axis'], float)
                if j['type'] == 'slide':
                    p = p + axis[d] * q[d]
                else:
                    Rk = _rot(axis[d], q[d]) @ Rk
                    p = anchor[d] - Rk @ jpos
                d += 1
            xpos[k], R[k] = p, Rk
        return dict(xpos=xpos, R=R, axis=axis, anchor=anchor)

    def _ancestor(self, dof, body):
        b = body
        while b > 0:
            if b == self.dof_body[dof]:
                return True
            b = self.bodies[b].parent
        return False

    def jac(self, kin, body, point):
        jp = np.zeros((3, self.nv)); jr = np.zeros((3, self.nv))
        for d in range(self.nv):
            if not self._ancestor(d, body):
                continue
            if self.dof_joint[d]['type'] == 'slide':
                jp[:, d] = kin['axis'][d]
            else:
                jp[:, d] = np.cross(kin['axis'][d], point - kin['anchor'][d])
                jr[:, d] = kin['axis'][d]
        return jp, jr

    def body_jacs(self, q):
        kin = self.kinematics(q)
        out = []
        for k in range(1, len(self.bodies)):
            if self.mass[k] == 0:
                continue
            com = kin['xpos'][k] + kin['R'][k] @ self.ipos[k]
            jp, jr = self.jac(kin, k, com)
            Iw = kin['R'][k] @ self.inertia[k] @ kin['R'][k].T
            out.append((self.mass[k], Iw, jp, jr))
        return out

    def mass_matrix(self, q):
        M = np.diag(self.armature).astype(float)
        for m, Iw, jp, jr in self.body_jacs(q):
            M = M + m * jp.T @ jp + jr.T @ Iw @ jr
        return M

    def bias(self, q, v, eps=1e-6):
        """velocity-product generalized force, with dJ/dt taken numerically along v"""
        J0 = self.body_jacs(q)
        Jp = self.body_jacs(q + eps * v)
        Jm = self.body_jacs(q - eps * v)
        c = np.zeros(self.nv)
        for (m, Iw, jp, jr), (_, _, jpp, jrp), (_, _, jpm, jrm) in zip(J0, Jp, Jm):
            ap = (jpp - jpm) / (2 * eps) @ v
            ar = (jrp - jrm) / (2 * eps) @ v
            w = jr @ v
            c += jp.T @ (m * ap) + jr.T @ (Iw @ ar + np.cross(w, Iw @ w))
        return c

    # -- constraint rows --------------------------------------------------------------------
    def _kbi(self, pos):
        tc = max(SOLREF[0], 2 * self.h)
        dmin, dmax, width, mid, power = SOLIMP
        b = 2 / (dmax * tc)
        k = 1 / (dmax * dmax * tc * tc * SOLREF[1] ** 2)
        x = abs(pos) / width
        y = (x ** power) / mid ** (power - 1) if x < mid else 1 - ((1 - x) ** power) / (1 - mid) ** (power - 1)
        imp = min(max(dmin + y * (dmax - dmin), dmin), dmax)
        if x > 1:
            imp = dmax
        return k, b, imp

    def rows(self, q, v):
        """list of (J row, aref, D) for the active limit and contact rows"""
        out = []
        for d, j in enumerate(self.dof_joint):
            if j['range'] is None:
                continue
            dlo, dhi = q[d] - j['range'][0], j['range'][1] - q[d]
            pos = min(dlo, dhi)
            if pos >= 0:
                continue
            sg = 1.0 if dlo < dhi else -1.0
            k, b, imp = self._kbi(pos)
            Jr = np.zeros(self.nv); Jr[d] = sg
            R = max(1e-15, (1 - imp) / imp * self.dof_invweight0[d])
            out.append((Jr, -b * (Jr @ v) - k * imp * pos, 1 / R))
        kin = self.kinematics(q)
        for body, gpos, r in self.foot:
            centre = kin['xpos'][body] + kin['R'][body] @ gpos
            dist = centre[2] - r                       # floor plane z = 0, normal +z
            pos = dist - self.margin
            if pos >= 0:
                continue
            cpos = centre - np.array([0, 0, r + 0.5 * dist])
            jp, _ = self.jac(kin, body, cpos)
            k, b, imp = self._kbi(pos)
            t = self.body_invweight0[body, 0]
            mu = self.friction
            invw = (t + mu * mu * t) * 2 * mu * mu / 1.0       # impratio 1
            R = max(1e-15, (1 - imp) / imp * invw)
            for tang in (np.array([0.0, 1.0, 0.0]), np.array([-1.0, 0.0, 0.0])):
                for s in (1.0, -1.0):
                    Jr = jp[2] + s * mu * (tang @ jp)
                    out.append((Jr, -b * (Jr @ v) - k * imp * pos, 1 / R))
        return out

    # -- one mjx.step ---------------------------------------------------------------------------
    def gravity_force(self, q):
        """generalized gravity force  sum_b m_b Jp_b' g  (part of MuJoCo's qfrc_bias, with the opposite sign)"""
        return sum(m * jp.T @ GRAVITY for m, _, jp, _ in self.body_jacs(q))

    def smooth_force(self, q, v, ctrl):
        tau = np.zeros(self.nv)
        nu = len(ctrl)
        tau[self.nv - nu:] = self.gear[self.nv - nu:] * np.clip(ctrl, -1, 1)
        return -self.bias(q, v) + self.gravity_force(q) - self.damping * v - self.stiffness * q + tau

    def solve(self, M, a0, rows):
        """min 1/2 (a-a0)' M (a-a0) + sum 1/2 D min(0, J a - aref)^2 : Newton with exact line search"""
        if not rows:
            return a0, np.zeros_like(a0)
        J = np.array([r[0] for r in rows]); ar = np.array([r[1] for r in rows]); D = np.array([r[2] for r in rows])

        def grad(a):
            res = J @ a - ar
            act = res < 0
            return M @ (a - a0) + J.T @ (D * act * res), act

        a = a0.copy()
        for it in range(100):
            g, act = grad(a)
            if np.linalg.norm(g) < 1e-11 * (1 + np.linalg.norm(M @ a0)):
                break
            Hm = M + (J.T * (D * act)) @ J
            p = -np.linalg.solve(Hm, g)
            # exact line search on the convex piecewise-quadratic: bisection on the derivative
            lo, hi = 0.0, 1.0
            while grad(a + hi * p)[0] @ p < 0:
                hi *= 2
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                if grad(a + mid * p)[0] @ p < 0:
                    lo = mid
                else:
                    hi = mid
            a = a + hi * p
        res = J @ a - ar
        force = -D * np.minimum(res, 0)
        return a, J.T @ force

    def step(self, q, v, ctrl):
        """returns (pose (x, y, cos, sin) of the robot body at the START state, qacc, q', v')"""
        q = np.asarray(q, float); v = np.asarray(v, float)
        M = self.mass_matrix(q)
        f = self.smooth_force(q, v, np.asarray(ctrl, float))
        a0 = np.linalg.solve(M, f)
        qacc, fc = self.solve(M, a0, self.rows(q, v))
        qint = np.linalg.solve(M + self.h * np.diag(self.damping), f + fc)
        v2 = v + self.h * qint
        q2 = q + self.h * v2
        kin = self.kinematics(q)
        pose = np.array([kin['xpos'][1][0], kin['xpos'][1][1], kin['R'][1][0, 0], kin['R'][1][1, 0]])
        return pose, qacc, q2, v2


class AntModel(TreeModel):
    def __init__(self):
        super().__init__(build_tables(), H, RHO, FRICTION, MARGIN, [GEAR] * 8)


def _rot(axis, angle):
    a = _unit(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def lagrangian_check(model, q, v, eps=1e-5):
    """bias from the Lagrangian  c = Mdot v - 1/2 d(v'Mv)/dq  with M differentiated numerically"""
    n = model.nv
    Mdot = (model.mass_matrix(q + eps * v) - model.mass_matrix(q - eps * v)) / (2 * eps)
    dT = np.zeros(n)
    for i in range(n):
        e = np.zeros(n); e[i] = eps
        dT[i] = 0.5 * (v @ model.mass_matrix(q + e) @ v - v @ model.mass_matrix(q - e) @ v) / (2 * eps)
    return Mdot @ v - dT


if __name__ == "__main__":
    mdl = AntModel()
    np.set_printoptions(precision=10, linewidth=160)
    print("masses", {b.name: m for b, m in zip(mdl.bodies, mdl.mass) if m})
    print("dof_invweight0", mdl.dof_invweight0)
    print("body_invweight0 ankle", mdl.body_invweight0[4])
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, 11); v = rng.uniform(-2, 2, 11)
    print("bias", mdl.bias(q, v))
    print("lagr", lagrangian_check(mdl, q, v))
