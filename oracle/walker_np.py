"""walker_np.py -- TEST INFRASTRUCTURE ONLY (float64, generic).

xmls/walker.xml restated as body/joint/geom tables for the generic tree model of oracle/ant_np.py:
planar base (x slide, z hinge with stiffness .1, body-y slide, :16-18), two legs of five hinges each --
hip_x / hip_z / hip_y in the thigh body (armature .01, damping 5, stiffness 10, :22-24,41-43), knee and
foot hinges about -y with offset anchors (default class: armature 1, damping 20, :5,27,30,46,49) --
capsule geoms of density 50, one foot sphere per leg (r = .02, the only geoms with contype, :32,51) against
the floor plane (margin 0, friction .75 from the default class), motors gear 10/10/30/30/40 with ctrlrange
+-1 (:99-111), timestep .02 (:9).  [derived] MuJoCo/MJX semantics, parity unpinned.
"""
import numpy as np

from .ant_np import TreeModel, Body, DEG

H = 0.02
RHO = 50.0
FRICTION = 0.75
MARGIN = 0.0
GEARS = [10.0, 10.0, 30.0, 30.0, 40.0]
Z0 = 0.42


def _hinge(axis, rng, armature=1.0, damping=20.0, stiffness=0.0, pos=(0, 0, 0)):
    a = np.asarray(axis, float)
    return dict(type='hinge', axis=a / np.linalg.norm(a), pos=pos, damping=damping, armature=armature,
                stiffness=stiffness, range=(rng[0] * DEG, rng[1] * DEG))


def build_tables():
    bodies = [Body('world', -1, (0, 0, 0))]
    robot = Body('robot', 0, (0, 0, Z0))
    robot.joints = [dict(type='slide', axis=(1, 0, 0), damping=10.0, armature=0.0, stiffness=0.0, range=None),
                    dict(type='hinge', axis=(0, 0, 1), damping=1.0, armature=0.0, stiffness=0.1, range=None),
                    dict(type='slide', axis=(0, 1, 0), damping=10.0, armature=0.0, stiffness=0.0, range=None)]
    robot.geoms = [dict(type='capsule', a=(0, 0.1, 0), b=(0, -0.1, 0), r=0.03, contact=False)]
    bodies.append(robot)
    for side, (ys, sx, sz) in (('right', (0.1, 1.0, 1.0)), ('left', (-0.1, -1.0, -1.0))):
        thigh = Body(side + '_thigh', 1, (0, ys, 0))
        hip = dict(armature=0.01, damping=5.0, stiffness=10.0)
        thigh.joints = [_hinge((sx, 0, 0), (-25, 5), **hip), _hinge((0, 0, sz), (-30, 35), **hip),
                        _hinge((0, 1, 0), (-100, 10), **hip)]
        thigh.geoms = [dict(type='capsule', a=(0, 0, 0), b=(0, 0, -0.2), r=0.03, contact=False)]
        bodies.append(thigh)
        it = len(bodies) - 1
        leg = Body(side + '_leg', it, (0, 0, -0.3))
        leg.joints = [_hinge((0, -1, 0), (-100, 0), pos=(0, 0, 0.1))]
        leg.geoms = [dict(type='capsule', a=(0, 0, 0.1), b=(0, 0, -0.1), r=0.02, contact=False)]
        bodies.append(leg)
        il = len(bodies) - 1
        foot = Body(side + '_foot', il, (0, 0, 0))
        foot.joints = [_hinge((0, -1, 0), (-45, 20), pos=(0, 0, -0.1))]
        foot.geoms = [dict(type='capsule', a=(0, 0, -0.1), b=(0.05, 0, -0.1), r=0.03, contact=False),
                      dict(type='sphere', pos=(0.05, 0, -0.1), r=0.02, contact=True)]
        bodies.append(foot)
    return bodies


class WalkerModel(TreeModel):
    def __init__(self):
        super().__init__(build_tables(), H, RHO, FRICTION, MARGIN, GEARS + GEARS)
