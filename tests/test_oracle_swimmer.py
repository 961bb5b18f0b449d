"""Swimmer (BASELINE config 3, articulated dynamics + joint-limit constraint rows):
closed-form checks of the C restatement and the cross-check against the independent float64
numpy formulation (oracle/gx_oracle_np.py).  [derived] MuJoCo semantics, parity unpinned."""
import numpy as np
import pytest

from helpers import task_config
from oracle import gx_oracle_np as onp

f32 = np.float32
SWIM = dict(robot_base='xmls/swimmer.xml')


def _engine(oracle, N, **kw):
    E = oracle.OracleEngine(task_config(N, **SWIM, **kw), n_candidates=6000)
    E.reset(check=False)
    return E


def _state(N, rng, limit_mode="mixed"):
    q = np.zeros((N, 5), f32)
    q[:, :2] = rng.uniform(-2, 2, (N, 2))
    q[:, 2] = rng.uniform(-6, 6, N)
    q[:, 3:] = rng.uniform(-1.6, 1.6, (N, 2))
    if limit_mode in ("mixed", "both"):
        k = N // 3 if limit_mode == "mixed" else N
        q[:k, 3] = rng.choice([-1, 1], k) * (1.7453293 + rng.uniform(1e-5, 0.05, k))
        q[k // 2:k, 4] = rng.choice([-1, 1], k - k // 2) * (1.7453293 + rng.uniform(1e-5, 0.05, k - k // 2))
    v = np.zeros((N, 5), f32)
    v[:, :2] = rng.uniform(-1, 1, (N, 2))
    v[:, 2:] = rng.uniform(-5, 5, (N, 3))
    pose0 = np.c_[q[:, :2], np.cos(q[:, 2]), np.sin(q[:, 2])].astype(f32)
    return dict(qpos=q, qvel=v, pose0=pose0, pose1=pose0[:, :2].copy(),
                objs=rng.uniform(-2, 2, (N, 9, 2)).astype(f32), done0=np.zeros(N, f32), done1=np.zeros(N, f32),
                steps=np.zeros(N, f32), key=np.array([1, 2], np.uint32), hist=2)


def test_dims_and_obs_layout(oracle):
    E = _engine(oracle, 4)
    assert (E.nq, E.nv, E.nu, E.na, E.D) == (5, 5, 2, 2, 46)   # obs 46
    s = _state(4, np.random.default_rng(0), "none")
    E.set_state(s)
    a = np.array([[0.3, -2.0]] * 4, f32)
    obs, r, d, info = E.step(a)
    np.testing.assert_array_equal(obs[:, 0:2], a)              # ctrl is the RAW action (no convert, no clip)
    assert info['qacc'].shape == (4, 5)
    st = E.get_state()
    np.testing.assert_array_equal(obs[:, 36:41], st['qpos'])
    np.testing.assert_array_equal(obs[:, 41:46], st['qvel'])


@pytest.mark.parametrize("mode", ["none", "mixed", "both"])
def test_step_c_vs_numpy(oracle, mode):
    N = 300
    E = _engine(oracle, N)
    rng = np.random.default_rng({"none": 0, "mixed": 1, "both": 2}[mode])
    s = _state(N, rng, mode)
    E.set_state(s)
    act = rng.uniform(-1.5, 1.5, (N, 2)).astype(f32)           # beyond ctrlrange: force is clipped
    obs, r, d, info = E.step(act)
    st = E.get_state()
    for i in range(N):
        pose, qacc, q2, v2 = onp.swimmer_step(s['qpos'][i], s['qvel'][i], act[i])
        np.testing.assert_allclose(info['qacc'][i], qacc, rtol=2e-4, atol=2e-3, err_msg=f"env {i}")
        np.testing.assert_allclose(st['qvel'][i], v2, rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose(st['qpos'][i], q2, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(st['pose0'][i], pose, atol=1e-6)


def test_momentum_and_symmetry(oracle):
    """no external force on the chain: the generalized momentum of the translation DOFs
    (incl. armature) is conserved by the unconstrained dynamics; mirrored state -> mirrored motion"""
    N = 64
    E = _engine(oracle, N)
    rng = np.random.default_rng(3)
    s = _state(N, rng, "none")
    s['qvel'][:, 2:] *= 0.2                     # gentle rates: the O(h^2) Euler drift stays tiny
    E.set_state(s)
    M0 = np.array([onp.swimmer_mass_bias(s['qpos'][i], s['qvel'][i])[0] for i in range(N)])
    p0 = np.einsum('nij,nj->ni', M0, s['qvel'].astype(float))[:, :2]
    act = np.zeros((N, 2), f32)
    E.step(act)
    st = E.get_state()
    M1 = np.array([onp.swimmer_mass_bias(st['qpos'][i], st['qvel'][i])[0] for i in range(N)])
    p1 = np.einsum('nij,nj->ni', M1, st['qvel'].astype(float))[:, :2]
    np.testing.assert_allclose(p1, p0, atol=2e-4)      # O(h^2) integrator drift only
    act = rng.uniform(-1, 1, (N, 2)).astype(f32)
    E.set_state(s)
    E.step(act)
    st = E.get_state()
    # mirror (y -> -y): angles, angular rates and torques flip sign
    sm = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in s.items()}
    for k in ('qpos', 'qvel'):
        sm[k][:, 1] *= -1; sm[k][:, 2:] *= -1
    sm['pose0'][:, 1] *= -1; sm['pose0'][:, 3] *= -1
    sm['objs'][..., 1] *= -1
    E.set_state(sm)
    E.step(-act)
    stm = E.get_state()
    np.testing.assert_allclose(stm['qpos'][:, 0], st['qpos'][:, 0], atol=2e-6)
    np.testing.assert_allclose(stm['qpos'][:, 1], -st['qpos'][:, 1], atol=2e-6)
    np.testing.assert_allclose(stm['qpos'][:, 2:], -st['qpos'][:, 2:], atol=2e-5)


def test_joint_limit_pushes_back(oracle):
    """a joint beyond +100 deg moving outward gets a restoring acceleration (limit row active);
    inside the range the same state gets none"""
    E = _engine(oracle, 2)
    s = _state(2, np.random.default_rng(0), "none")
    s['qpos'][:, 2:] = 0; s['qvel'][:] = 0
    s['qpos'][0, 3] = 1.78; s['qpos'][1, 3] = 1.70
    s['qvel'][:, 3] = 2.0
    E.set_state(s)
    _, _, _, info = E.step(np.zeros((2, 2), f32))
    assert info['qacc'][0, 3] < -50 and abs(info['qacc'][1, 3]) < 5


def test_episode_runs_and_stays_finite(oracle):
    N, T = 50, 150
    E = oracle.OracleEngine(task_config(N, seed=4, num_steps=T, **SWIM), n_candidates=30000)
    E.reset()
    rng = np.random.RandomState(0)
    hit = 0
    for t in range(T):
        obs, r, d, info = E.step(rng.uniform(-1, 1, (N, 2)).astype(f32))
        assert np.isfinite(obs).all()
        hit += int((np.abs(E.get_state()['qpos'][:, 3:]) > 1.7453293).sum())
        E.reset_done()
    assert hit > 0                        # the limit rows were exercised
    q = E.get_state()['qpos']
    # soft limits (timeconst 0.06) against a 20 N.m motor: steady penetration up to ~0.67 rad
    assert (np.abs(q[:, 3:]) < 3.0).all()    # (dynamic overshoot included)
