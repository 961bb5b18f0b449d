"""Pins the PRNG restatement on PUBLISHED known answers (the only reference-side
pin that exists for this path), then cross-checks the C and the
numpy transcriptions of jax.random.{split,uniform,randint}."""
import numpy as np
import pytest

from oracle import gx_oracle_np as onp

# Random123 / jax tests/random_test.py::testThreefry2x32 known answers
KAT = [
    ((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6b200159, 0x99ba4efe)),
    ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
    ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0)),
]


@pytest.mark.parametrize("key,ctr,want", KAT)
def test_threefry_known_answers(oracle, key, ctr, want):
    assert oracle.threefry2x32(*key, *ctr) == want
    y0, y1 = onp.threefry2x32(key[0], key[1], [ctr[0]], [ctr[1]])
    assert (int(y0[0]), int(y1[0])) == want


def test_split_known_answer(oracle):
    # JAX documentation (PRNG design note): random.split(random.PRNGKey(0))
    want = np.array([[4146024105, 967050713], [2718843009, 1272950319]], np.uint32)
    np.testing.assert_array_equal(oracle.split((0, 0), 2), want)
    np.testing.assert_array_equal(onp.split(np.array([0, 0], np.uint32), 2), want)


def test_uniform_known_answers_via_published_normals(oracle):
    """jax.random.normal(key, (1,)) = sqrt(2) * erfinv(uniform(key, (1,), minval=nextafter(-1, 0), maxval=1))
    [jax/_src/random.py _normal_real].  Published values (JAX docs, "Sharp bits" / PRNG design):
        random.normal(PRNGKey(0), (1,))            -> -0.20584226
        key, subkey = split(PRNGKey(0)); random.normal(subkey, (1,)) -> -1.2515389
    They pin the bits->float mapping and the size-1 counter layout of the `uniform` restatement."""
    from scipy.special import erfinv
    lo, hi = float(np.nextafter(np.float32(-1), np.float32(0))), 1.0
    for key, want in (((0, 0), -0.20584226), (tuple(oracle.split((0, 0), 2)[1]), -1.2515389)):
        for u in (oracle.uniform(key, lo, hi), float(onp.uniform(np.array(key, np.uint32), lo, hi))):
            got = np.sqrt(2.0) * erfinv(np.float64(u))
            assert abs(got - want) < 3e-7, (key, u, got, want)


@pytest.mark.parametrize("n", [1, 2, 3, 10, 1001])
def test_split_c_vs_numpy(oracle, n):
    rng = np.random.default_rng(n)
    for _ in range(3):
        key = rng.integers(0, 2**32, 2, dtype=np.uint32)
        np.testing.assert_array_equal(oracle.split(key, n), onp.split(key, n))


def test_uniform_c_vs_numpy_and_range(oracle):
    rng = np.random.default_rng(0)
    vals = []
    for _ in range(2000):
        key = rng.integers(0, 2**32, 2, dtype=np.uint32)
        lo, hi = -1.6, 1.6
        a = oracle.uniform(key, lo, hi)
        b = float(onp.uniform(key, lo, hi))
        assert a == b
        assert np.float32(lo) <= a < np.float32(hi)
        vals.append(a)
    vals = np.array(vals)
    assert abs(vals.mean()) < 0.08 and abs(vals.std() - 3.2 / np.sqrt(12)) < 0.05


@pytest.mark.parametrize("n,span", [(1, 7), (4, 414), (5, 20731), (2000, 20731), (2001, 65537), (16, 1)])
def test_randint_c_vs_numpy(oracle, n, span):
    rng = np.random.default_rng(n + span)
    key = rng.integers(0, 2**32, 2, dtype=np.uint32)
    a = oracle.randint(key, n, span)
    b = onp.randint(key, n, span)
    np.testing.assert_array_equal(a, b)
    assert a.min() >= 0 and a.max() < span


def test_key_chain_is_first_child_of_split(oracle):
    """update_data: key, _ = split(key, 2) (engine.py:431) -- the engine's key after t steps."""
    E = oracle.OracleEngine({'env_num': 2, '_seed': 5}, n_candidates=4000)
    E.reset(check=False)
    key = np.array([0, 5], np.uint32)
    for _ in range(4):
        E.step(np.zeros((2, 2), np.float32))
        key = onp.split(key, 2)[0]
        np.testing.assert_array_equal(E.get_state()['key'], key)
