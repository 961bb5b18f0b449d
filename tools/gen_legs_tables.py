#!/usr/bin/env python3
"""Table generator for the generic "planar base + legs of serial hinges" dynamics (Walker): derives every
number from the robot MJCF with tools/mjcf_model.py (float64, MuJoCo compile rules) and prints the table in
the two spellings the sources carry -- a C initialiser for oracle/gx_oracle_legs.inc and constexpr arrays for
guardx_amd/csrc/gx_robot_legs.h.

    python tools/gen_legs_tables.py /path/to/xmls/walker.xml {c|hip|py}

`tables(xml)` returns the numbers as a dict (tests/test_model_constants.py compares the carried tables with it
whenever the MJCF files are present).  [derived]: see tools/mjcf_model.py.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mjcf_model import Model  # noqa: E402

SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
SOLREF = (0.02, 1.0)


def tables(xml):
    m = Model(xml)
    robot = m.body('robot')
    # legs = chains of bodies hanging off the robot body, in MJCF order
    legs = []
    for k, b in enumerate(m.bodies):
        if b['parent'] == robot:
            chain = [k]
            while True:
                kids = [c for c, bb in enumerate(m.bodies) if bb['parent'] == chain[-1]]
                if not kids:
                    break
                chain.append(kids[0])
            legs.append(chain)
    q0 = np.zeros(m.nv)
    kin = m.kinematics(q0)
    origin = kin['xpos'][robot]
    dof0 = 3
    T = dict(nleg=len(legs), h=m.timestep, z0=m.bodies[robot]['pos'][2], mtot=float(m.mass.sum()),
             mB=float(m.mass[robot]), IB=float(m.inertia[robot][2, 2]),
             base_damp=[j['damping'] for j in m.bodies[robot]['joints']],
             base_stiff=m.bodies[robot]['joints'][1]['stiffness'], legs=[])
    assert np.allclose(m.ipos[robot], 0), "base COM offset not supported"
    tc = max(SOLREF[0], 2 * m.timestep)
    T['kK'] = 1 / (SOLIMP[1] ** 2 * tc ** 2 * SOLREF[1] ** 2)
    T['kB'] = 2 / (SOLIMP[1] * tc)
    floor = m.bodies[0]['geoms'][0]
    d = dof0
    for chain in legs:
        L = dict(axis=[], dp=[], damp=[], arm=[], stiff=[], lo=[], hi=[], gear=[], invw=[], bodies=[])
        prev_anchor = np.zeros(3)
        link = -1
        anchors = []
        for k in chain:
            b = m.bodies[k]
            for j in b['joints']:
                link += 1
                anchor = kin['anchor'][d] - origin
                L['axis'].append((j['axis'] / np.linalg.norm(j['axis'])).tolist())
                L['dp'].append((anchor - prev_anchor).tolist())
                prev_anchor = anchor
                anchors.append(anchor)
                L['damp'].append(j['damping']); L['arm'].append(j['armature']); L['stiff'].append(j['stiffness'])
                L['lo'].append(float(j['range'][0])); L['hi'].append(float(j['range'][1]))
                act = [a for a in m.actuators if a['joint'] == j['name']][0]
                L['gear'].append(act['gear'])
                L['invw'].append(float(m.dof_invweight0[d]))
                d += 1
            if m.mass[k] > 0:
                com = kin['xpos'][k] + m.ipos[k] - origin
                I = m.inertia[k]
                L['bodies'].append(dict(link=link, m=float(m.mass[k]), c=(com - anchors[link]).tolist(),
                                        I=[I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]))
            for g in b['geoms']:
                if g['contype']:
                    centre = kin['xpos'][k] + g['pos'] - origin
                    mu = max(g['friction'][0], floor['friction'][0])
                    t = m.body_invweight0[k, 0]
                    L['foot'] = dict(link=link, s=(centre - anchors[link]).tolist(), r=float(g['size'][0]),
                                     margin=max(g['margin'], floor['margin']), mu=mu,
                                     invw_pyr=float((t + mu * mu * t) * 2 * mu * mu))
        T['legs'].append(L)
    T['K'] = len(T['legs'][0]['axis'])
    T['nb'] = len(T['legs'][0]['bodies'])
    return T


def _f(x):
    if x == 0:
        return "0.0f"
    t = "%.9g" % x
    if "." not in t and "e" not in t:
        t += ".0"
    return t + "f"


def _arr(v):
    return "{" + ", ".join(_arr(x) if isinstance(x, (list, tuple)) else _f(x) for x in v) + "}"


def emit(T, style):
    legs = T['legs']
    foot = legs[0]['foot']
    rows = [
        ("axis", [L['axis'] for L in legs]), ("dp", [L['dp'] for L in legs]),
        ("damp", [L['damp'] for L in legs]), ("arm", [L['arm'] for L in legs]), ("stiff", [L['stiff'] for L in legs]),
        ("lo", [L['lo'] for L in legs]), ("hi", [L['hi'] for L in legs]), ("gear", [L['gear'] for L in legs]),
        ("invw", [L['invw'] for L in legs]),
        ("bm", [[b['m'] for b in L['bodies']] for L in legs]), ("bc", [[b['c'] for b in L['bodies']] for L in legs]),
        ("bI", [[b['I'] for b in L['bodies']] for L in legs]), ("fs", [L['foot']['s'] for L in legs]),
    ]
    scal = [("h", T['h']), ("z0", T['z0']), ("mtot", T['mtot']), ("mB", T['mB']), ("IB", T['IB']),
            ("dbx", T['base_damp'][0]), ("dbt", T['base_damp'][1]), ("dby", T['base_damp'][2]), ("kt", T['base_stiff']),
            ("fr", foot['r']), ("margin", foot['margin']), ("mu", foot['mu']), ("invw_pyr", foot['invw_pyr']),
            ("kK", T['kK']), ("kB", T['kB'])]
    blink = [[b['link'] for b in L['bodies']] for L in legs]
    out = []
    if style == 'c':
        out.append("/* generated by tools/gen_legs_tables.py from xmls/walker.xml */")
        out.append("    .nleg = %d, .K = %d, .nb = %d, .flink = %d," % (T['nleg'], T['K'], T['nb'], foot['link']))
        out.append("    " + " ".join(".%s = %s," % (k, _f(v)) for k, v in scal))
        out.append("    .blink = {" + ", ".join("{" + ", ".join(str(x) for x in r) + "}" for r in blink) + "},")
        for k, v in rows:
            out.append("    .%s = %s," % (k, _arr(v)))
    else:
        out.append("    // generated by tools/gen_legs_tables.py from xmls/walker.xml")
        out.append("    static constexpr int kLegs = %d, kK = %d, kNb = %d, kFlink = %d;" % (T['nleg'], T['K'], T['nb'], foot['link']))
        for k, v in scal:
            out.append("    static constexpr float c_%s = %s;" % (k, _f(v)))
        out.append("    static constexpr int c_blink[%d][%d] = {%s};" % (T['nleg'], T['nb'], ", ".join("{" + ", ".join(str(x) for x in r) + "}" for r in blink)))
        for k, v in rows:
            a = np.array(v)
            dims = "".join("[%d]" % n for n in a.shape)
            out.append("    static constexpr float c_%s%s = %s;" % (k, dims, _arr(v)))
    return "\n".join(out)


if __name__ == "__main__":
    T = tables(sys.argv[1])
    style = sys.argv[2] if len(sys.argv) > 2 else 'py'
    if style == 'py':
        import json
        print(json.dumps(T, indent=1))
    else:
        print(emit(T, style))
