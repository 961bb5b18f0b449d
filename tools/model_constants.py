#!/usr/bin/env python3
"""Model constants of the robots the HIP path implements, derived in float64 the way the
MuJoCo compiler derives them from the robot MJCF (removes hand-copied
numbers as a source of silent error).  The geometric inputs are restated here with the
reference line they come from; the printed values are what oracle/gx_oracle.c and
guardx_amd/csrc/gx_robot.h carry.

[derived]: MuJoCo's geom mass/inertia formulas (user_objects.cc SetInertia), mj_setConst's
dof_invweight0 = diag(M(qpos0)^-1), degrees->radians of joint ranges.
"""
import numpy as np

pi = np.pi


def sphere(r, rho):
    m = rho * 4 / 3 * pi * r ** 3
    return m, 0.4 * m * r * r


def box(hx, hy, hz, rho):
    m = rho * 8 * hx * hy * hz
    return m, m / 3 * (hx * hx + hy * hy)          # about z


def capsule_perp(r, length, rho):
    """capsule of cylinder length `length`: mass and inertia about an axis perpendicular to it"""
    vc, vs = pi * r * r * length, 4 / 3 * pi * r ** 3
    m = rho * (vc + vs)
    mc, ms = rho * vc, rho * vs
    i = mc * (3 * r * r + length * length) / 12
    i += 2 * ms * r * r / 5 + ms * length * (3 * r + 2 * length) / 8
    return m, i


def point():
    # xmls/point.xml:5 density 1; :19 sphere r=.1 at origin; :20 box half .05 at (.1,0,0)
    ms, Is = sphere(0.1, 1.0)
    mb, Ib = box(0.05, 0.05, 0.05, 1.0)
    m = ms + mb
    mxc = mb * 0.1
    Io = Is + Ib + mb * 0.1 ** 2
    print("POINT  m=%.17g  m*xc=%.17g  Io=%.17g" % (m, mxc, Io))
    print("       h=0.02 (point.xml:3)  damping .01 .01 .005 (:16-18)  gear .3 (:37-39)")
    print("       1/m=%.17g  1/(m+h*d)=%.17g  Io+h*dz=%.17g" % (1 / m, 1 / (m + 0.02 * 0.01), Io + 0.02 * 0.005))


def swimmer():
    # xmls/swimmer.xml:3 timestep .03; :6 armature .1; :18,23,27 capsules r=.02 density 1000,
    # fromto (.3..15), (0..-.15), (0..-.15); bodies at (0,0,.03), (.15,0,0), (-.15,0,0);
    # :24,28 hinge range +-100 deg; :58-59 motors gear 20 ctrlrange +-1
    m, Ic = capsule_perp(0.02, 0.15, 1000.0)
    arm = 0.1
    a = np.array([[0.225, 0.0, 0.0],
                  [0.15, -0.075, 0.0],
                  [0.15, -0.15, -0.075]])       # COM_i = p + sum_k a[i,k] u(alpha_k)
    # M at qpos0 (all angles 0): n_k = (0, 1)
    g = np.zeros((3, 3, 2))
    n = np.array([0.0, 1.0])
    for i in range(3):
        for j in range(3):
            g[i, j] = sum(a[i, k] * n for k in range(j, 3))
    M = np.zeros((5, 5))
    M[0, 0] = M[1, 1] = 3 * m
    for j in range(3):
        M[0, 2 + j] = M[2 + j, 0] = m * sum(g[i, j, 0] for i in range(3))
        M[1, 2 + j] = M[2 + j, 1] = m * sum(g[i, j, 1] for i in range(3))
        for k in range(3):
            M[2 + j, 2 + k] = m * sum(g[i, j] @ g[i, k] for i in range(3)) + Ic * (3 - max(j, k))
    M += arm * np.eye(5)
    A = np.linalg.inv(M)
    lim = 100 * pi / 180
    print("SWIMMER link mass=%.17g  I_perp=%.17g  armature=%.3g  total mass+arm=%.17g" % (m, Ic, arm, 3 * m + arm))
    print("        a =", a.tolist())
    print("        dof_invweight0[motor1_rot]=%.17g  [motor2_rot]=%.17g" % (A[3, 3], A[4, 4]))
    print("        range=+-%.17g rad   h=0.03  gear=20" % lim)
    dmax, tc = 0.95, max(0.02, 2 * 0.03)
    print("        solref: timeconst=max(.02, 2h)=%.3g -> b=%.17g k=%.17g ; solimp=(.9,.95,.001,.5,2)"
          % (tc, 2 / (dmax * tc), 1 / (dmax * dmax * tc * tc)))


if __name__ == "__main__":
    point()
    swimmer()
