"""C restatement vs the independent numpy restatement (oracle/gx_oracle_np.py)."""
import numpy as np
import pytest

from helpers import task_config, random_state
from oracle import gx_oracle_np as onp

f32 = np.float32


def _lidar_close(a, b, pos, bins, tol=2e-5):
    """edge-aware lidar comparison: an angle within 1e-5 bins of a bin edge may legally land
    on either side (the bin index is the discontinuous part)."""
    frac = np.abs(pos - np.round(pos))
    risky = (frac < 1e-4).any(axis=1)
    np.testing.assert_allclose(a[~risky], b[~risky], rtol=0, atol=tol)
    return int(risky.sum())


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_step_c_vs_numpy(oracle, seed):
    N = 512
    cfg = task_config(N, seed=seed, num_steps=200)
    E = oracle.OracleEngine(cfg, n_candidates=4000)
    E.reset(check=False)
    rng = np.random.default_rng(seed)
    s = random_state(N, 8, rng)
    s['hist'] = [2, 1, 0][seed]
    E.set_state(s)
    act = rng.uniform(-1, 1, (N, 2)).astype(f32)
    obs_c, r_c, d_c, info = E.step(act)
    obs_n, r_n, d_n, cost_n, new, (pg, ph), qacc_n = onp.step(s, act, cfg)
    # non-lidar columns
    cols = list(range(0, 5)) + list(range(37, 43))
    np.testing.assert_allclose(obs_c[:, cols], obs_n[:, cols], rtol=2e-5, atol=2e-5)
    risky = _lidar_close(obs_c[:, 5:21], obs_n[:, 5:21], pg, 16) + \
        _lidar_close(obs_c[:, 21:37], obs_n[:, 21:37], ph, 16)
    assert risky < N // 20
    np.testing.assert_allclose(r_c, r_n, rtol=0, atol=2e-6)
    np.testing.assert_allclose(info['cost'], cost_n, rtol=0, atol=2e-6)
    np.testing.assert_allclose(info['qacc'], qacc_n, rtol=3e-5, atol=1e-3)
    # masks: identical except where a distance sits within float noise of its threshold
    dg = np.linalg.norm(s['objs'][:, 0] - s['qpos'][:, :2], axis=1)
    safe = np.abs(dg - 0.5) > 1e-5
    np.testing.assert_array_equal(d_c[safe], d_n[safe])
    st = E.get_state()
    np.testing.assert_allclose(st['qpos'], new['qpos'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(st['qvel'], new['qvel'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(st['pose0'], new['pose0'], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(st['steps'][safe], new['steps'][safe])


def test_sample_layout_c_vs_numpy(oracle):
    """first valid layouts of a reset: the C sampler against the scalar numpy transcription"""
    M = 3000
    E = oracle.OracleEngine(task_config(4, seed=7), n_candidates=M)
    E.reset(check=False)
    pool_c = E.get_pool()
    key = np.array([0, 7], np.uint32)
    keys = onp.split(key, M)
    got = []
    for j in range(M):
        lay, ok = onp.sample_layout(keys[j])
        if ok:
            got.append(lay)
        if len(got) == 6:
            break
    assert len(got) >= 3
    np.testing.assert_array_equal(np.stack(got), pool_c[:len(got)])


def test_sample_layout_with_pillars_c_vs_numpy(oracle):
    """synthetic config-5 objects: the pillars are drawn after the hazards with their own keepout"""
    M = 1500
    ext = [-3, -3, 3, 3]
    E = oracle.OracleEngine(task_config(4, seed=9, pillars_num=5, hazards_num=4, placements_extents=ext),
                            n_candidates=M)
    E.reset(check=False)
    pool_c = E.get_pool()
    assert pool_c.shape[1:] == (1 + 4 + 5 + 1, 2)
    keys = onp.split(np.array([0, 9], np.uint32), M)
    got = []
    for j in range(M):
        lay, ok = onp.sample_layout(keys[j], hazards_num=4, extents=ext, pillars_num=5)
        if ok:
            got.append(lay)
        if len(got) == 5:
            break
    assert len(got) >= 3
    np.testing.assert_array_equal(np.stack(got), pool_c[:len(got)])


def test_get_layout_indices_c_vs_numpy(oracle):
    N = 37
    E = oracle.OracleEngine(task_config(N, seed=11), n_candidates=8000)
    E.reset(check=False)
    L = E.layout_size
    pool = E.get_pool()
    idx = onp.randint(np.array([0, 11], np.uint32), N, L)
    st = E.get_state()
    np.testing.assert_array_equal(st['objs'], pool[idx][:, :9])
    np.testing.assert_array_equal(st['qpos'][:, :2], pool[idx][:, 9])
