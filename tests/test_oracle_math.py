"""Accuracy of the deterministic fp32 sin/cos/atan2/exp shared (by construction, not by
code) between the checker and the kernels, against double-precision libm."""
import numpy as np


def test_sincos_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-400, 400, 400000), rng.uniform(-np.pi, np.pi, 200000),
                        np.linspace(-1e4, 1e4, 100001)]).astype(np.float32)
    s, c, _, _ = oracle.math_probe(x, np.zeros_like(x))
    xd = x.astype(np.float64)
    assert np.abs(s - np.sin(xd)).max() < 1.3e-7
    assert np.abs(c - np.cos(xd)).max() < 1.3e-7
    # |x| up to 1e5 still within 1e-6 (Cody-Waite with 3 constants)
    xb = rng.uniform(-1e5, 1e5, 100000).astype(np.float32)
    sb, cb, _, _ = oracle.math_probe(xb, np.zeros_like(xb))
    assert np.abs(sb - np.sin(xb.astype(np.float64))).max() < 1e-6


def test_sincos_special(oracle):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan], np.float32)
    s, c, _, _ = oracle.math_probe(x, np.zeros_like(x))
    assert s[0] == 0 and c[0] == 1 and s[1] == 0 and c[1] == 1
    assert np.isnan(s[2:]).all() and np.isnan(c[2:]).all()


def test_atan2_accuracy_and_quadrants(oracle):
    rng = np.random.default_rng(1)
    x = rng.uniform(-5, 5, 600000).astype(np.float32)
    y = rng.uniform(-5, 5, 600000).astype(np.float32)
    _, _, a, _ = oracle.math_probe(x, y)
    assert np.abs(a - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 3.6e-7
    xs = np.array([1, -1, 0, 0, -1, -1, 0.0, -0.0, np.inf, -np.inf, np.nan], np.float32)
    ys = np.array([0, 0, 1, -1, 0.0, -0.0, 0.0, 0.0, np.inf, np.inf, 1.0], np.float32)
    _, _, a, _ = oracle.math_probe(xs, ys)
    want = np.arctan2(ys.astype(np.float64), xs.astype(np.float64))
    np.testing.assert_allclose(a[:-1], want[:-1], atol=3e-7)
    assert np.isnan(a[-1])
    assert np.signbit(a[5]) and abs(a[5] + np.pi) < 3e-7      # atan2(-0, -1) = -pi


def test_exp_accuracy(oracle):
    x = np.linspace(-87, 5, 500001).astype(np.float32)
    _, _, _, e = oracle.math_probe(x, np.zeros_like(x))
    t = np.exp(x.astype(np.float64))
    assert (np.abs(e - t) / t).max() < 1.6e-7
    xs = np.array([0.0, -87.5, -1000.0, -np.inf, np.nan], np.float32)
    _, _, _, e = oracle.math_probe(xs, np.zeros_like(xs))
    assert e[0] == 1 and e[1] == 0 and e[2] == 0 and e[3] == 0 and np.isnan(e[4])
