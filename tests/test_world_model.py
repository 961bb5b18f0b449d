"""SURVEY.md row f4, the `World.build` half -- and the ONE artefact of this path the reference checkout holds.

`/root/reference/safe_rl_libX/result.xml` is the MJCF `World.build` itself wrote (world.py:331-332) for
Goal_Point_8Hazards.  tools/world_model.py restates `Engine.build_world_config` (engine.py:335-384) and `World.build`
(world.py:104-326) without xmltodict / mujoco; here the assembled tree must equal that file element by element -- tags,
order, attribute names, attribute STRINGS -- and the index tables derived from it (engine.py:302-316) must be what the
kernels and the CPU checker hard-code.  The file is read in place (it does not travel and is never copied): the tests
that need it skip where /root/reference is absent."""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
XML_DIR = "/root/reference/safe_rl_envs/safe_rl_envs/xmls"
RESULT_XML = "/root/reference/safe_rl_libX/result.xml"
needs_ref = pytest.mark.skipif(not (os.path.isdir(XML_DIR) and os.path.isfile(RESULT_XML)),
                               reason="reference checkout (robot MJCF files, result.xml) not present")

import world_model as wm  # noqa: E402
from guardx_amd import configuration  # noqa: E402


def _ref():
    return wm.Node.from_element(ET.parse(RESULT_XML).getroot())


@needs_ref
def test_assembled_world_equals_the_reference_result_xml():
    root, _ = wm.assemble(dict(configuration("Goal_Point_8Hazards")), XML_DIR)
    ref = _ref()
    # top-level sections in the reference's order, then everything below them
    assert [c.tag for c in root.flat()] == [c.tag for c in ref.flat()] == \
        ['size', 'option', 'default', 'worldbody', 'sensor', 'actuator', 'equality', 'asset']
    for mine, theirs in zip(root.flat(), ref.flat()):
        assert mine.signature() == theirs.signature(), mine.tag
    assert root.signature() == ref.signature()


@needs_ref
def test_the_facts_the_hot_path_rests_on_are_in_the_reference_file():
    """read off the reference's own result.xml (not off the restatement): what DESIGN.md sections 0.1 / 2 assume"""
    ref = _ref()
    assert ref.one('option').attrib == {'timestep': '0.02'}
    # <default>: ONE class; <motor> THEN <velocity> write its single actuator default, in this order
    assert [c.tag for c in ref.one('default').flat()] == ['geom', 'joint', 'motor', 'velocity', 'site']
    for tag in ('motor', 'velocity'):
        assert ref.one('default').one(tag).attrib == {'ctrlrange': '-1 1', 'ctrllimited': 'true',
                                                      'forcerange': '-.05 .05', 'forcelimited': 'true'}
    acts = ref.one('actuator').flat()
    assert [(a.tag, a.attrib['gear'], a.attrib['joint']) for a in acts] == \
        [('general', '0.3', 'robot_x'), ('general', '0.3', 'robot_y'), ('general', '0.3', 'robot_z')]
    assert all(set(a.attrib) == {'gear', 'joint', 'name'} for a in acts)       # no gain / bias / limit of their own
    bodies = ref.one('worldbody').all('body')
    assert [b.attrib['name'] for b in bodies] == ['robot', 'goal'] + [f'hazard{i}' for i in range(8)]   # robot FIRST
    rb = bodies[0]
    assert [(j.attrib['type'], j.attrib['axis'], j.attrib['name'], j.attrib['damping']) for j in rb.all('joint')] == \
        [('slide', '1 0 0', 'robot_x', '0.01'), ('slide', '0 1 0', 'robot_y', '0.01'), ('hinge', '0 0 1', 'robot_z', '0.005')]
    assert float(rb.attrib['pos'].split()[2]) == 0.1 and rb.attrib['quat'] == '1.0 0.0 0.0 0.0'
    for b in bodies[1:]:
        haz = b.attrib['name'] != 'goal'
        assert float(b.attrib['pos'].split()[2]) == (0.02 if haz else 0.0)
        js = b.all('joint')                                                    # two damped, unlimited slides: static unless pushed
        assert [(j.attrib['type'], j.attrib['axis'], j.attrib['damping'], j.attrib['limited']) for j in js] == \
            [('slide', '1 0 0', '1', 'false'), ('slide', '0 1 0', '1', 'false')]
        g = b.one('geom')
        assert g.attrib['contype'] == '0' and g.attrib['conaffinity'] == '0'   # never collide
        assert g.attrib['type'] == ('cylinder' if haz else 'sphere')
        assert g.attrib['size'] == ('0.3 0.01' if haz else '0.5')
    floor = ref.one('worldbody').one('geom')
    assert floor.attrib['name'] == 'floor' and floor.attrib['type'] == 'plane' and 'margin' not in floor.attrib
    # the robot's sphere rests exactly ON the floor: dist = z - r = 0 with margin 0 -> inactive in MJX
    sphere = next(g for g in rb.all('geom') if g.attrib['name'] == 'robot')
    assert np.float32(rb.attrib['pos'].split()[2]) - np.float32(sphere.attrib['size']) == np.float32(0.0)


@needs_ref
def test_index_tables_equal_what_kernels_and_checker_hard_code(oracle):
    """engine.py:302-316 on the assembled model of every robot: the robot body is body 1 and its joints come first
    (qpos[:nq] / qvel[:nv] of the world arrays are the robot's: engine.py:760-766), robot_x / robot_y sit where
    R::place (gx_robot*.h) and the checker's layout2qpos (oracle/gx_oracle.c, engine.py:635-638) write the layout's
    robot position, the goal / hazard bodies follow in layout order."""
    from guardx_amd.engine import _ROBOTS
    from helpers import task_config
    import re
    expect_xy = {'xmls/point.xml': (0, 1), 'xmls/swimmer.xml': (0, 1), 'xmls/ant.xml': (0, 2), 'xmls/walker.xml': (0, 2)}
    hdr = {'xmls/point.xml': ('gx_robot.h', 'PointRobotT'), 'xmls/swimmer.xml': ('gx_robot.h', 'SwimmerRobot'),
           'xmls/ant.xml': ('gx_robot_ant.h', 'AntRobot'), 'xmls/walker.xml': ('gx_robot_legs.h', 'WalkerRobot')}
    for base, (rid, nq, nv, nu, z, dt, act) in _ROBOTS.items():
        cfg = dict(configuration("Goal_Point_8Hazards"), robot_base=base)
        root, t = wm.assemble(cfg, XML_DIR)
        assert t['body_name2xpos_id'] == {'robot': 1, 'goal': t['bodies'].index('goal'),
                                          'hazards': [t['bodies'].index(f'hazard{i}') for i in range(8)]}
        names = list(t['joint_name2qpos_id'])
        assert (wm.robot_dims(XML_DIR, base)) == (nq, nv, nu), base
        assert wm.robot_z_height(wm.Node.parse(open(os.path.join(XML_DIR, os.path.basename(base))).read())) == z
        assert float(root.one('option').attrib['timestep']) == dt
        # the robot's joints occupy qpos[0:nq]; every static body adds two slides behind them, goal first
        assert all(t['joint_name2qpos_id'][n] < nq for n in names[:names.index('goal_x')])
        assert t['joint_name2qpos_id']['goal_x'] == nq and t['joint_name2qpos_id']['hazard7_y'] == nq + 2 * 9 - 1
        assert t['nq'] == nq + 2 * 9
        ix, iy = t['joint_name2qpos_id']['robot_x'], t['joint_name2qpos_id']['robot_y']
        assert (ix, iy) == expect_xy[base], base
        # ... the checker places the robot there (reset from a one-layout pool: qpos is zero except robot_x / robot_y)
        E = oracle.OracleEngine(task_config(8, seed=2, robot_base=base), n_candidates=3000)
        E.reset(check=False)
        q = E.get_state()['qpos']
        pool = E.get_pool()
        rows = {tuple(r[-1]) for r in pool}
        for e in range(8):
            nz = np.nonzero(q[e])[0].tolist()
            assert set(nz) <= {ix, iy}, (base, nz)
            assert (float(q[e, ix]), float(q[e, iy])) in {(float(a), float(b)) for a, b in rows}
        # ... and so does the HIP trait: `place` writes q[ix] = rx, q[iy] = ry
        fn, struct = hdr[base]
        src = open(os.path.join(ROOT, "guardx_amd/csrc", fn)).read()
        body = src[src.index("struct " + struct):]
        m = re.search(r'static void place\(float \(&q\)\[NQ\], float rx, float ry\)\s*\{\s*q\[(\d+)\] = rx; q\[(\d+)\] = ry;', body)
        assert m and (int(m.group(1)), int(m.group(2))) == (ix, iy), (base, m and m.groups())


def test_build_world_config_quirks():
    """engine.py:370-384: with hazards_num == 0 the function falls off its end (the return is inside the `if`) -- the
    Engine then raises TypeError, as guardx_amd.Engine does; robot_rot None -> random_rot() == 0.0"""
    assert wm.build_world_config({'hazards_num': 0}) is None
    wc = wm.build_world_config({'hazards_num': 2, 'hazards_size': 0.25, 'goal_size': 0.4, 'robot_rot': None})
    assert wc['robot_rot'] == 0.0 and list(wc['geoms']) == ['goal', 'hazard0', 'hazard1']
    assert wm.convert(wc['geoms']['hazard1']['size']) == '0.25 0.01' and wm.convert(wc['geoms']['goal']['rgba']) == '0.0 1.0 0.0 0.25'
    assert 'floor_size' not in wc
    assert wm.build_world_config({'hazards_num': 1, 'floor_display_mode': True})['floor_size'] == [2.1, 2.1, 1]


def test_node_groups_same_tag_siblings_like_xmltodict():
    n = wm.Node.parse('<a x="1"><b i="0"/><c/><b i="1"/><!-- gone --><d/></a>')
    assert [(c.tag, dict(c.attrib)) for c in n.flat()] == [('b', {'i': '0'}), ('b', {'i': '1'}), ('c', {}), ('d', {})]
    n.children['c'] = [n.one('c'), wm.Node('c', {'new': '1'})]     # assigning an existing key keeps its position
    n.children['z'] = [wm.Node('z')]
    assert [c.tag for c in n.flat()] == ['b', 'b', 'c', 'c', 'd', 'z']
    assert wm.Node.parse(n.to_xml()).signature() == n.signature()
