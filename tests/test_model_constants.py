"""the model constants carried by the HIP headers and by the CPU checker equal what
tools/model_constants.py derives from the robot MJCF files (when those are available: they live in the
reference checkout, which exists in the build container only), and the two carriers agree with each other
and with the hand-restated tables of oracle/ant_np.py."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XML_DIR = "/root/reference/safe_rl_envs/safe_rl_envs/xmls"
NUM = r'(-?[0-9]+\.?[0-9]*(?:[eE][-+]?[0-9]+)?)f'


def _struct_text(path, name):
    s = open(path).read()
    i = s.index("struct " + name)
    j = s.find("\nstruct ", i + 1)
    return s[i:j if j > 0 else len(s)]


def _hip_constants(path, struct):
    txt = _struct_text(path, struct)
    return {k: float(v) for k, v in re.findall(r'\b(k[A-Za-z0-9]+|A[1-3][1-3])\s*=\s*' + NUM, txt)}


def _c_defines(path, prefix):
    s = open(path).read()
    return {k: float(v) for k, v in re.findall(r'#define\s+' + prefix + r'_(\w+)\s+\(?' + NUM, s)}


HIP_POINT = _hip_constants(os.path.join(ROOT, "guardx_amd/csrc/gx_robot.h"), "PointRobot")
HIP_SWIM = _hip_constants(os.path.join(ROOT, "guardx_amd/csrc/gx_robot.h"), "SwimmerRobot")
HIP_ANT = _hip_constants(os.path.join(ROOT, "guardx_amd/csrc/gx_robot_ant.h"), "AntRobot")
C_ANT = _c_defines(os.path.join(ROOT, "oracle/gx_oracle_ant.inc"), "AN")
C_SWIM = _c_defines(os.path.join(ROOT, "oracle/gx_oracle.c"), "SW")
C_POINT = _c_defines(os.path.join(ROOT, "oracle/gx_oracle.c"), "PT")

ANT_NAMES = {'H': 'kH', 'A': 'kA', 'A2': 'kA2', 'L': 'kL', 'RF': 'kRf', 'Z0': 'kZ0', 'MARGIN': 'kMargin', 'MU': 'kMu',
             'MB': 'kMB', 'IB': 'kIB', 'MA': 'kMA', 'ITA': 'kITA', 'MK': 'kMK', 'LC': 'kLC', 'ITK': 'kITK',
             'DIK': 'kDIK', 'MTOT': 'kMtot', 'LBB': 'kLbb', 'INVW_HIP': 'kInvwHip', 'INVW_ANK': 'kInvwAnk',
             'INVW_PYR': 'kInvwPyr', 'K': 'kK', 'B': 'kB', 'LIM30': 'kLim30', 'LIM70': 'kLim70', 'GEAR': 'kGear', 'GK': 'kGK'}
SWIM_NAMES = {'H': 'kH', 'M': 'kM', 'IC': 'kIc', 'ARM': 'kArm', 'GEAR': 'kGear', 'LIM': 'kLim', 'INVW2': 'kInvW2',
              'INVW3': 'kInvW3', 'K': 'kK', 'B': 'kB', 'A11': 'A11', 'A21': 'A21', 'A22': 'A22', 'A31': 'A31',
              'A32': 'A32', 'A33': 'A33'}
POINT_NAMES = {'H': 'kH', 'MXC': 'kMxc', 'IO': 'kIo', 'DXY': 'kDxy', 'DT': 'kDt', 'GEAR': 'kGear', 'CTRLLIM': 'kCtrlLim',
               'FORCELIM': 'kForceLim', 'KV': 'kKv'}


def _same_f32(a, b):
    return np.float32(a) == np.float32(b)


def test_hip_and_checker_carry_the_same_constants():
    for names, c, h in ((ANT_NAMES, C_ANT, HIP_ANT), (SWIM_NAMES, C_SWIM, HIP_SWIM), (POINT_NAMES, C_POINT, HIP_POINT)):
        for cn, hn in names.items():
            assert cn in c and hn in h, (cn, hn)
            assert _same_f32(c[cn], h[hn]), (cn, c[cn], h[hn])
    assert len(HIP_ANT) >= len(ANT_NAMES)


def test_ant_constants_match_the_hand_restated_tables():
    sys.path.insert(0, ROOT)
    from oracle import ant_np
    m = ant_np.AntModel()
    names = [b.name for b in m.bodies]
    ank, aux = names.index('ankle_1'), names.index('aux_1')
    assert _same_f32(HIP_ANT['kMK'], m.mass[ank]) and _same_f32(HIP_ANT['kMA'], m.mass[aux])
    assert _same_f32(HIP_ANT['kMtot'], m.mass.sum())
    assert _same_f32(HIP_ANT['kInvwHip'], m.dof_invweight0[3]) and _same_f32(HIP_ANT['kInvwAnk'], m.dof_invweight0[4])
    t, mu = m.body_invweight0[ank, 0], ant_np.FRICTION
    assert _same_f32(HIP_ANT['kInvwPyr'], (t + mu * mu * t) * 2 * mu * mu)
    assert _same_f32(HIP_ANT['kLC'], np.linalg.norm(m.ipos[ank]))


@pytest.mark.skipif(not os.path.isdir(XML_DIR), reason="robot MJCF files (reference checkout) not present")
def test_headers_match_the_mjcf_files():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import model_constants
    derived = model_constants.constants(XML_DIR)
    for robot, hip in (("point", HIP_POINT), ("swimmer", HIP_SWIM), ("ant", HIP_ANT)):
        checked = 0
        for name, val in derived[robot].items():
            if name in hip:
                assert abs(hip[name] - val) <= 2e-7 * abs(val) + 1e-30, (robot, name, hip[name], val)
                checked += 1
        assert checked >= {"point": 8, "swimmer": 14, "ant": 27}[robot], (robot, checked)
    # Point: the mass enters through literal expressions
    txt = _struct_text(os.path.join(ROOT, "guardx_amd/csrc/gx_robot.h"), "PointRobot")
    assert "0.005188790204786391" in txt and abs(0.005188790204786391 - derived['point']['kM']) < 1e-17


@pytest.mark.skipif(not os.path.isdir(XML_DIR), reason="robot MJCF files (reference checkout) not present")
def test_parser_agrees_with_hand_tables_on_the_ant():
    sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
    from mjcf_model import Model
    from oracle import ant_np
    pm, hm = Model(os.path.join(XML_DIR, "ant.xml")), ant_np.AntModel()
    np.testing.assert_allclose(pm.mass, hm.mass, rtol=1e-14)
    np.testing.assert_allclose(pm.dof_invweight0, hm.dof_invweight0, rtol=1e-12)
    np.testing.assert_allclose(pm.body_invweight0, hm.body_invweight0, rtol=1e-12)
    q = np.random.default_rng(0).uniform(-1, 1, 11)
    np.testing.assert_allclose(pm.mass_matrix(q), hm.mass_matrix(q), rtol=1e-12, atol=1e-18)
    assert pm.timestep == ant_np.H


@pytest.mark.skipif(not os.path.isdir(XML_DIR), reason="robot MJCF files (reference checkout) not present")
def test_walker_tables_match_the_mjcf_file():
    """the generated tables carried by gx_robot_legs.h / gx_oracle_legs.inc are what the generator derives now,
    and the generator's parser agrees with the hand-restated tables of oracle/walker_np.py"""
    sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
    import gen_legs_tables
    from mjcf_model import Model
    from oracle import walker_np
    T = gen_legs_tables.tables(os.path.join(XML_DIR, "walker.xml"))
    hip = open(os.path.join(ROOT, "guardx_amd/csrc/gx_robot_legs.h")).read()
    cin = open(os.path.join(ROOT, "oracle/gx_oracle_legs.inc")).read()
    for line in gen_legs_tables.emit(T, 'hip').splitlines()[1:]:
        assert line.strip() in hip, line[:80]
    for line in gen_legs_tables.emit(T, 'c').splitlines()[1:]:
        assert line.strip() in cin, line[:80]
    pm, hm = Model(os.path.join(XML_DIR, "walker.xml")), walker_np.WalkerModel()
    np.testing.assert_allclose(pm.mass, hm.mass, rtol=1e-14)
    np.testing.assert_allclose(pm.dof_invweight0, hm.dof_invweight0, rtol=1e-10)
    q = np.random.default_rng(0).uniform(-0.5, 0.5, 13)
    np.testing.assert_allclose(pm.mass_matrix(q), hm.mass_matrix(q), rtol=1e-10, atol=1e-16)


SYNTH_MJCF = """<mujoco>
  <default>
    <joint damping="0.5"/>
    <motor ctrlrange="-2 2" ctrllimited="true" forcerange="-3 3" forcelimited="true" gear="7"/>
    <position kp="4" ctrlrange="-1 1"/>
  </default>
  <worldbody>
    <body name="b" pos="0 0 1">
      <joint name="j0" type="slide" axis="1 0 0"/>
      <joint name="j1" type="hinge" axis="0 0 1"/>
      <geom type="sphere" size="0.1"/>
    </body>
  </worldbody>
  <actuator>
    <general joint="j0" gear="0.5"/>
    <motor joint="j1"/>
    <velocity joint="j1" kv="9" forcelimited="false"/>
    <general joint="j0" biastype="none" gainprm="2"/>
  </actuator>
</mujoco>"""


def test_actuator_default_class_resolution(tmp_path):
    """MuJoCo keeps ONE actuator default per class; the shortcut children of <default> write it in
    document order and <general> inherits whatever it does not set (XML reference, default/motor ..
    default/velocity) -- the rule that decides point.xml's actuators.  Synthetic file, written here."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mjcf_model import Model
    f = tmp_path / "synth.xml"
    f.write_text(SYNTH_MJCF)
    m = Model(str(f))
    d = m.actuator_default          # <motor> wrote the limits and gear, <position> then gain/bias and ctrlrange
    assert d['gear'] == 7 and d['ctrllimited'] is True and list(d['ctrlrange']) == [-1, 1]
    assert d['forcelimited'] is True and list(d['forcerange']) == [-3, 3]
    assert d['biastype'] == 'affine' and list(d['biasprm']) == [0, -4, 0] and d['gainprm'][0] == 4
    g, mo, ve, g2 = m.actuators
    assert g['gear'] == 0.5 and g['biastype'] == 'affine' and list(g['biasprm']) == [0, -4, 0] and g['gainprm'][0] == 4
    assert g['ctrllimited'] is True and g['forcelimited'] is True and list(g['forcerange']) == [-3, 3]
    assert mo['gear'] == 7 and mo['biastype'] == 'none' and mo['gainprm'][0] == 1 and list(mo['biasprm']) == [0, 0, 0]
    assert ve['biastype'] == 'affine' and list(ve['biasprm']) == [0, 0, -9] and ve['gainprm'][0] == 9
    assert ve['forcelimited'] is False and ve['ctrllimited'] is True
    assert g2['biastype'] == 'none' and g2['gainprm'][0] == 2 and list(g2['biasprm']) == [0, -4, 0]


@pytest.mark.skipif(not os.path.isdir(XML_DIR), reason="robot MJCF files (reference checkout) not present")
def test_point_actuators_resolve_to_the_limited_velocity_servo():
    """point.xml:4-10,37-39: the <general gear=.3> actuators inherit ctrlrange +-1, forcerange +-.05 and
    the affine bias (0, 0, -1) of <velocity>; Engine.action_space follows actuator_ctrllimited
    (engine.py:291-297).  The other robots' <motor> actuators carry no bias and no force limit."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mjcf_model import Model
    m = Model(os.path.join(XML_DIR, "point.xml"))
    assert len(m.actuators) == 3
    for a in m.actuators:
        assert a['kind'] == 'general' and a['gear'] == 0.3
        assert a['ctrllimited'] is True and list(a['ctrlrange']) == [-1, 1]
        assert a['forcelimited'] is True and list(a['forcerange']) == [-0.05, 0.05]
        assert a['gaintype'] == 'fixed' and a['gainprm'][0] == 1
        assert a['biastype'] == 'affine' and list(a['biasprm']) == [0, 0, -1]
    from guardx_amd.engine import _ROBOTS
    for xml, nu in (("point.xml", 2), ("swimmer.xml", 2), ("ant.xml", 8), ("walker.xml", 10)):
        mm = Model(os.path.join(XML_DIR, xml))
        lo, hi, dim = _ROBOTS['xmls/' + xml][6]
        assert dim == nu
        for a in mm.actuators[:nu]:
            assert a['ctrllimited'] is True and (lo, hi) == tuple(a['ctrlrange'])
        if xml != "point.xml":
            assert all(a['biastype'] == 'none' and a['forcelimited'] is False for a in mm.actuators)
