#!/usr/bin/env python3
"""Constants carried by guardx_amd/csrc/gx_robot.h / gx_robot_ant.h (and by the CPU checker), derived in
float64 from the robot MJCF files themselves by tools/mjcf_model.py (no hand-copied
numbers, no `mujoco`).

    python tools/model_constants.py /path/to/safe_rl_envs/safe_rl_envs/xmls

`constants(xml_dir)` returns {robot: {name: value}} with the names used in the headers;
tests/test_model_constants.py checks the headers against it whenever the MJCF files are available.
[derived]: MuJoCo compile rules and mj_setConst, see tools/mjcf_model.py.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mjcf_model import Model  # noqa: E402

SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)      # MuJoCo defaults (no solimp/solref attribute in the robot files)
SOLREF = (0.02, 1.0)


def _kb(timestep):
    tc = max(SOLREF[0], 2 * timestep)       # refsafe
    return 1 / (SOLIMP[1] ** 2 * tc ** 2 * SOLREF[1] ** 2), 2 / (SOLIMP[1] * tc)


def point(m):
    r = m.body('robot')
    mass = m.mass[r]
    # planar: x, y slides + hinge about z through the body origin
    io = m.inertia[r][2, 2] + mass * (m.ipos[r][0] ** 2 + m.ipos[r][1] ** 2)
    d = [j['damping'] for j in m.dof_joint]
    h = m.timestep
    a = m.actuators[0]
    # the three <general> actuators resolve identically (same class default, same attributes)
    assert all(_same_actuator(a, b) for b in m.actuators[1:])
    assert a['ctrllimited'] is True and a['forcelimited'] is True and a['gaintype'] == 'fixed' and a['biastype'] == 'affine'
    assert a['gainprm'][0] == 1.0 and a['biasprm'][0] == 0.0 and a['biasprm'][1] == 0.0
    return dict(kH=h, kM=mass, kMxc=mass * m.ipos[r][0], kIo=io, kDxy=d[0], kDt=d[2], kGear=a['gear'],
                kInvM=1 / mass, kInvA=1 / (mass + h * d[0]), kEi=io + h * d[2],
                kCtrlLim=a['ctrlrange'][1], kForceLim=a['forcerange'][1], kKv=-a['biasprm'][2])


def _same_actuator(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ('gear', 'ctrllimited', 'forcelimited', 'ctrlrange', 'forcerange',
                                                    'gaintype', 'biastype', 'gainprm', 'biasprm'))


def _plain_motor(a):
    """<motor ctrllimited ctrlrange=+-1>: fixed gain 1, no bias, no force limit"""
    return (a['gaintype'] == 'fixed' and a['biastype'] == 'none' and a['gainprm'][0] == 1.0 and a['ctrllimited'] is True
            and list(a['ctrlrange']) == [-1.0, 1.0] and a['forcelimited'] is False)


def swimmer(m):
    links = [m.body('robot'), m.body('mid'), m.body('back')]
    k, b = _kb(m.timestep)
    assert all(_plain_motor(a) for a in m.actuators)
    out = dict(kH=m.timestep, kM=m.mass[links[0]], kIc=m.inertia[links[0]][2, 2], kArm=m.dof_joint[0]['armature'],
               kGear=m.actuators[0]['gear'], kLim=m.dof_joint[3]['range'][1],
               kInvW2=m.dof_invweight0[3], kInvW3=m.dof_invweight0[4], kK=k, kB=b)
    # COM_i = p + sum_k a[i][k] u(alpha_k): body offsets and geom centres along the link axes
    out.update(A11=m.ipos[links[0]][0], A21=m.bodies[links[1]]['pos'][0], A22=m.ipos[links[1]][0],
               A31=m.bodies[links[1]]['pos'][0], A32=m.bodies[links[2]]['pos'][0], A33=m.ipos[links[2]][0])
    out['kImu'] = 1 / (3 * out['kM'] + out['kArm'])
    return out


def ant(m):
    robot, leg, aux = m.body('robot'), m.body('front_left_leg'), m.body('aux_1')
    ank = aux + 1                                           # the unnamed ankle body
    a = np.linalg.norm(m.bodies[aux]['pos'])
    foot = [g for g in m.bodies[ank]['geoms'] if g['contype']][0]
    length = np.linalg.norm(foot['pos'])
    lc = np.linalg.norm(m.ipos[ank])
    u = m.ipos[ank] / lc
    iak = u @ m.inertia[ank] @ u
    itk = m.inertia[ank][2, 2]
    mb = m.mass[robot] + 4 * m.mass[leg]
    ib = m.inertia[robot][2, 2] + 4 * (m.inertia[leg][2, 2] + m.mass[leg] * (m.ipos[leg][0] ** 2 + m.ipos[leg][1] ** 2))
    k, b = _kb(m.timestep)
    mu = max(foot['friction'][0], m.bodies[0]['geoms'][0]['friction'][0])
    t = m.body_invweight0[ank, 0]
    hip, ankle = m.dof_joint[3], m.dof_joint[4]
    assert all(_plain_motor(x) for x in m.actuators)
    return dict(kH=m.timestep, kA=a, kA2=a / 2, kL=length, kRf=foot['size'][0], kZ0=m.bodies[robot]['pos'][2],
                kMargin=max(foot['margin'], m.bodies[0]['geoms'][0]['margin']), kMu=mu,
                kMB=mb, kIB=ib, kMA=m.mass[aux], kITA=m.inertia[aux][2, 2], kMK=m.mass[ank], kLC=lc,
                kITK=itk, kDIK=iak - itk, kMtot=m.mass.sum(), kLbb=m.mass[ank] * lc * lc + itk + ankle['armature'],
                kInvwHip=m.dof_invweight0[3], kInvwAnk=m.dof_invweight0[4],
                kInvwPyr=(t + mu * mu * t) * 2 * mu * mu,       # impratio 1
                kK=k, kB=b, kLim30=hip['range'][1], kLim70=ankle['range'][1], kGear=m.actuators[0]['gear'],
                kD7=abs(m.bodies[aux]['pos'][0]) / a, kGK=m.mass[ank] * 9.81 * lc)   # default gravity


def constants(xml_dir):
    return dict(point=point(Model(os.path.join(xml_dir, 'point.xml'))),
                swimmer=swimmer(Model(os.path.join(xml_dir, 'swimmer.xml'))),
                ant=ant(Model(os.path.join(xml_dir, 'ant.xml'))))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    for robot, vals in constants(sys.argv[1]).items():
        print(robot.upper())
        for k, v in vals.items():
            print("    %-10s %.17g" % (k, v))
