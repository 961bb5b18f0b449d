"""Walker (xmls/walker.xml): the generic "planar base + serial-hinge legs" fp32 C restatement against the
independent float64 tree model (oracle/walker_np.py on the machinery of oracle/ant_np.py), plus closed-form
checks.  [derived] MuJoCo/MJX semantics, parity unpinned."""
import numpy as np
import pytest

from helpers import task_config, WALKER, WALKER_LO, WALKER_HI
from oracle import walker_np, ant_np

f32 = np.float32


@pytest.fixture(scope="module")
def model():
    return walker_np.WalkerModel()


def _rand_qv(rng, th=1.0, beyond=0.15):
    q = np.zeros(13)
    q[0] = rng.uniform(-2, 2); q[1] = rng.uniform(-th, th); q[2] = rng.uniform(-2, 2)
    w = (WALKER_HI - WALKER_LO).astype(float)
    q[3:] = WALKER_LO - beyond * w + rng.uniform(0, 1 + 2 * beyond, 10) * w
    if rng.random() < 0.5:      # nearly straight legs with the feet pitched down: the foot spheres reach the floor
        q[3:] *= 0.15
        q[7] = rng.uniform(-0.8, -0.1); q[12] = rng.uniform(-0.8, -0.1)
    v = rng.normal(0, 2, 13) * rng.choice([0.0, 0.3, 1.0])
    return q.astype(f32).astype(float), v.astype(f32).astype(float)


def test_generic_model_self_consistency(model):
    rng = np.random.default_rng(0)
    for _ in range(4):
        q, v = _rand_qv(rng)
        M = model.mass_matrix(q)
        np.testing.assert_allclose(M, M.T, atol=1e-15)
        assert np.linalg.eigvalsh(M).min() > 0
        np.testing.assert_allclose(model.bias(q, v), ant_np.lagrangian_check(model, q, v), rtol=5e-5, atol=5e-7)


def test_dynamics_c_vs_generic(oracle, model):
    rng = np.random.default_rng(1)
    for _ in range(80):
        q, v = _rand_qv(rng, th=3.0)
        ctrl = rng.uniform(-1.5, 1.5, 10).astype(f32).astype(float)
        _, _, _, pose, Mc, fc = oracle.walker_probe(q, v, ctrl)
        M = model.mass_matrix(q)
        f = model.smooth_force(q, v, ctrl)
        assert np.abs(Mc - M).max() < 5e-7 * np.abs(M).max()
        assert np.abs(fc - f).max() < 2e-6 * (1 + np.abs(f).max())
        kin = model.kinematics(q)
        np.testing.assert_allclose(pose, [kin['xpos'][1][0], kin['xpos'][1][1], np.cos(q[1]), np.sin(q[1])], atol=2e-6)


def test_step_c_vs_generic(oracle, model):
    """full mjx.step incl. limit and contact rows; headings within 1 rad of zero (the x slide and the body-y
    slide are parallel at |th| = pi/2)"""
    rng = np.random.default_rng(2)
    nlim = ncon = 0
    for _ in range(120):
        q, v = _rand_qv(rng, th=1.0)
        ctrl = rng.uniform(-1.5, 1.5, 10).astype(f32).astype(float)
        q2, v2, qacc, pose, _, _ = oracle.walker_probe(q, v, ctrl)
        pose_r, qacc_r, q2_r, v2_r = model.step(q, v, ctrl)
        rows = model.rows(q, v)
        nlim += sum(1 for r in rows if np.count_nonzero(r[0]) == 1)
        ncon += sum(1 for r in rows if np.count_nonzero(r[0]) > 1)
        assert np.abs(qacc - qacc_r).max() < 3e-3 * (1 + np.abs(qacc_r).max())
        assert np.abs(v2 - v2_r).max() < 3e-3 * (1 + np.abs(v2_r).max())
        assert np.abs(q2 - q2_r).max() < 3e-3 * (1 + np.abs(q2_r).max())
    assert nlim > 100 and ncon > 100     # the sample exercises both kinds of rows


def test_gravity_and_springs_at_rest(oracle, model):
    """at qpos0 the only generalized force is the weight of the forward-offset feet about the pitch joints;
    a thigh swung forward about hip_y is pulled back by gravity and by the hip spring"""
    q = np.zeros(13); v = np.zeros(13)
    _, _, _, _, _, f0 = oracle.walker_probe(q, v, np.zeros(10))
    np.testing.assert_allclose(f0, model.gravity_force(q), atol=1e-8)
    assert np.all(f0[[3, 4, 8, 9]] == 0) and np.all(np.abs(f0[[5, 6, 7, 10, 11, 12]]) > 1e-3)
    q[5] = -0.6                                  # right hip_y
    _, _, _, _, _, f = oracle.walker_probe(q, v, np.zeros(10))
    assert f[5] > 0.6 * 10 * 0.9                 # spring 10 N m/rad
    np.testing.assert_allclose(f[5] - 10 * 0.6, model.gravity_force(q)[5], rtol=1e-4)
    assert model.gravity_force(q)[5] > 0


def test_foot_contact_rows(model):
    """pitching the foot down drives the foot sphere through the floor: 4 pyramid rows per foot"""
    q = np.zeros(13); q[7] = -0.6; q[12] = -0.6
    rows = model.rows(q, np.zeros(13))
    assert sum(1 for r in rows if np.count_nonzero(r[0]) > 1) == 8


def test_engine_dims_and_obs_layout(oracle):
    E = oracle.OracleEngine(task_config(4, **WALKER), n_candidates=6000)
    obs = E.reset(check=False)
    assert (E.nq, E.nv, E.nu, E.na, E.D) == (13, 13, 10, 10, 70)
    st = E.get_state()
    np.testing.assert_array_equal(obs[:, 44:57], st['qpos'])
    assert np.all(st['qpos'][:, [1] + list(range(3, 13))] == 0)
    a = np.random.default_rng(0).uniform(-2, 2, (4, 10)).astype(f32)
    obs, r, d, info = E.step(a)
    np.testing.assert_array_equal(obs[:, 0:10], a)
    assert info['qacc'].shape == (4, 13)
    st = E.get_state()
    np.testing.assert_array_equal(obs[:, 44:57], st['qpos'])
    np.testing.assert_array_equal(obs[:, 57:70], st['qvel'])
