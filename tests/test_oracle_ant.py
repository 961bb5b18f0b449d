"""Ant (xmls/ant.xml): the closed-form fp32 C restatement against the independent generic float64
model (oracle/ant_np.py: body/joint tables, MuJoCo's kinematics rule, point Jacobians, numerically
differentiated velocity products, dense Newton solve with exact line search) plus closed-form
checks.  [derived] MuJoCo/MJX semantics, parity unpinned."""
import numpy as np
import pytest

from helpers import task_config, ANT, ANT_SIGMA
from oracle import ant_np

f32 = np.float32


@pytest.fixture(scope="module")
def model():
    return ant_np.AntModel()


def _rand_qv(rng, th=1.0, vel=2.0):
    q = np.zeros(11)
    q[0] = rng.uniform(-2, 2); q[1] = rng.uniform(-th, th); q[2] = rng.uniform(-2, 2)
    for leg in range(4):
        q[3 + 2 * leg] = rng.uniform(-0.9, 0.9)
        q[4 + 2 * leg] = ANT_SIGMA[leg] * rng.uniform(0.2, 1.5)
    v = rng.normal(0, vel, 11) * rng.choice([0.0, 0.3, 1.0])
    return q.astype(f32).astype(float), v.astype(f32).astype(float)


def test_generic_model_self_consistency(model):
    """the generic model's own bias (dJ/dt numerically) equals the Lagrangian form Mdot v - dT/dq;
    M is symmetric positive definite; total mass on the translation block"""
    rng = np.random.default_rng(0)
    for _ in range(5):
        q, v = _rand_qv(rng)
        M = model.mass_matrix(q)
        np.testing.assert_allclose(M, M.T, atol=1e-15)
        assert np.linalg.eigvalsh(M).min() > 0
        assert abs(M[0, 0] - model.mass.sum()) < 1e-15 and abs(M[2, 2] - model.mass.sum()) < 1e-15
        np.testing.assert_allclose(model.bias(q, v), ant_np.lagrangian_check(model, q, v), rtol=2e-5, atol=2e-8)


def test_compiled_constants(model):
    """MuJoCo compile rules: masses (density 5), joint ranges in radians, setconst inverse weights"""
    m = {b.name: x for b, x in zip(model.bodies, model.mass)}
    assert abs(m['robot'] - 5 * 4 / 3 * np.pi * 0.06 ** 3) < 1e-15
    cap = lambda L: 5 * (np.pi * 0.02 ** 2 * L + 4 / 3 * np.pi * 0.02 ** 3)
    assert abs(m['aux_1'] - cap(0.05 * np.sqrt(2))) < 1e-15
    assert abs(m['ankle_3'] - (cap(0.1 * np.sqrt(2)) + 5 * 4 / 3 * np.pi * 0.02 ** 3)) < 1e-15
    # armature 1 dominates the leg DOFs; the base is light
    assert np.all(np.abs(model.dof_invweight0[3:] - 1.0) < 1e-4)
    assert abs(model.dof_invweight0[0] * model.mass.sum() - 1) < 1e-4      # couplings to the legs are tiny


def test_dynamics_c_vs_generic(oracle, model):
    """mass matrix and smooth force of the closed form against the generic model"""
    rng = np.random.default_rng(1)
    for _ in range(100):
        q, v = _rand_qv(rng, th=3.0)
        ctrl = rng.uniform(-1.5, 1.5, 8).astype(f32).astype(float)
        _, _, _, pose, Mc, fc = oracle.ant_probe(q, v, ctrl)
        M = model.mass_matrix(q)
        f = model.smooth_force(q, v, ctrl)
        assert np.abs(Mc - M).max() < 3e-7 * np.abs(M).max()
        assert np.abs(fc - f).max() < 1e-6 * (1 + np.abs(f).max())
        kin = model.kinematics(q)
        np.testing.assert_allclose(pose, [kin['xpos'][1][0], kin['xpos'][1][1], np.cos(q[1]), np.sin(q[1])], atol=2e-6)


def test_step_c_vs_generic(oracle, model):
    """one full mjx.step (limit + contact rows, Newton solve, implicit-damping Euler).  The x slide and
    the body-y slide become parallel at |th| = pi/2 (mass matrix singular), so headings stay within 1 rad;
    fp32 then carries ~1e-3 relative error on the light base (cond(M) ~ 1e4)."""
    rng = np.random.default_rng(2)
    nrows = 0
    for _ in range(150):
        q, v = _rand_qv(rng, th=1.0)
        ctrl = rng.uniform(-1.5, 1.5, 8).astype(f32).astype(float)
        q2, v2, qacc, pose, _, _ = oracle.ant_probe(q, v, ctrl)
        pose_r, qacc_r, q2_r, v2_r = model.step(q, v, ctrl)
        nrows += len(model.rows(q, v))
        assert np.abs(qacc - qacc_r).max() < 2e-3 * (1 + np.abs(qacc_r).max())
        assert np.abs(v2 - v2_r).max() < 2e-3 * (1 + np.abs(v2_r).max())
        assert np.abs(q2 - q2_r).max() < 2e-3 * (1 + np.abs(q2_r).max())
    assert nrows > 500          # the sample does exercise limits and contacts


def test_rest_pose_and_first_step(oracle):
    """qpos0 has every ankle 30 deg outside its range: the limit rows push all four feet down by the same
    amount (sign of the ankle axes), nothing else moves, no contact yet"""
    q = np.zeros(11, f32); q[0], q[2] = 0.7, -0.4
    q2, v2, qacc, pose, _, _ = oracle.ant_probe(q, np.zeros(11, f32), np.zeros(8, f32))
    np.testing.assert_array_equal(pose, np.array([0.7, -0.4, 1.0, 0.0], f32))
    ank = q2[4::2] * ANT_SIGMA
    assert np.all(ank > 0.05) and np.ptp(ank) < 1e-6
    assert np.abs(q2[3::2]).max() < 1e-6 and abs(q2[1]) < 1e-6
    assert np.abs(q2[[0, 2]] - q[[0, 2]]).max() < 1e-5


def test_ctrl_is_clipped_for_the_force_only(oracle):
    rng = np.random.default_rng(3)
    q, v = _rand_qv(rng)
    a = oracle.ant_probe(q, v, np.full(8, 1.0, f32))
    b = oracle.ant_probe(q, v, np.full(8, 7.0, f32))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_foot_contact_pushes_back(oracle, model):
    """ankle driven past ~58 deg puts the foot sphere inside the floor margin: the contact rows are
    active and the normal force opposes further pitching (ankle acceleration below the free value)"""
    q = np.zeros(11); v = np.zeros(11)
    for leg in range(4):
        q[4 + 2 * leg] = ANT_SIGMA[leg] * 1.15
    rows = model.rows(q, v)
    assert len(rows) == 16                      # 4 feet x 4 pyramid edges, no limit row
    _, _, qacc, _, M, f = oracle.ant_probe(q, v, np.zeros(8))
    free = np.linalg.solve(M.astype(float), f.astype(float))
    assert np.all((qacc[4::2] - free[4::2]) * ANT_SIGMA < 0)


def test_engine_dims_and_obs_layout(oracle):
    E = oracle.OracleEngine(task_config(4, **ANT), n_candidates=6000)
    obs = E.reset(check=False)
    assert (E.nq, E.nv, E.nu, E.na, E.D) == (11, 11, 8, 8, 64)    # obs 64
    st = E.get_state()
    np.testing.assert_array_equal(obs[:, 42:53], st['qpos'])
    assert np.all(st['qpos'][:, [1] + list(range(3, 11))] == 0)    # layout2qpos: only robot_x / robot_y
    np.testing.assert_array_equal(obs[:, 53:64], 0)
    a = np.random.default_rng(0).uniform(-2, 2, (4, 8)).astype(f32)
    obs, r, d, info = E.step(a)
    np.testing.assert_array_equal(obs[:, 0:8], a)                  # ctrl is the RAW action
    assert info['qacc'].shape == (4, 11)
    st = E.get_state()
    np.testing.assert_array_equal(obs[:, 42:53], st['qpos'])
    np.testing.assert_array_equal(obs[:, 53:64], st['qvel'])


def test_reset_done_obs_comes_from_the_fake_step(oracle):
    """mjx_reset_done (engine.py:719-729): the obs of a re-initialised env is built from data stepped once
    from the reset qpos -- for the ant that step moves the ankles -- while the stored qpos stays at rest"""
    E = oracle.OracleEngine(task_config(8, goal_size=5.0, **ANT), n_candidates=6000)   # every env done at once
    E.reset(check=False)
    a = np.zeros((8, 8), f32)
    E.step(a)
    obs, r, d, info = E.step(a)
    assert d.all()
    o2 = E.reset_done()
    st = E.get_state()
    assert np.all(st['qpos'][:, 3:] == 0) and np.all(st['qvel'] == 0)
    assert np.all(np.abs(o2[:, 42 + 4:53:2]) > 0.05)               # ankles moved in the obs
    np.testing.assert_array_equal(o2[:, 42], st['qpos'][:, 0])     # base x unchanged by the fake step (to fp32)


def test_gravity_pulls_the_feet_down(oracle, model):
    """legs inside their ranges, at rest, no ctrl: the only generalized force is the weight of the ankle
    links about the ankle joints (MK g LC cos(beta) per unit armature) -- every foot accelerates downwards"""
    q = np.zeros(11)
    for leg in range(4):
        q[4 + 2 * leg] = ANT_SIGMA[leg] * 0.8
    assert not model.rows(q, np.zeros(11))
    _, _, qacc, _, _, f = oracle.ant_probe(q, np.zeros(11), np.zeros(8))
    expect = 0.0012236798040145848 * 9.81 * 0.080392694440424323 * np.cos(0.8)
    np.testing.assert_allclose(f[4::2] * ANT_SIGMA, expect, rtol=1e-5)
    assert np.all(qacc[4::2] * ANT_SIGMA > 0.9 * expect / 1.00002)
    np.testing.assert_allclose(model.gravity_force(q)[4::2] * ANT_SIGMA, expect, rtol=1e-7)
