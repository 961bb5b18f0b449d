#!/usr/bin/env python3
"""Fit the polynomial coefficients used by the deterministic fp32 math of the
GUARD step path (gx_sincos / gx_atan2 / gx_exp_neg).

The oracle (oracle/gx_oracle.c) and the HIP kernels (guardx_amd/csrc) each carry
their own transcription of the constants printed here; both evaluate them with
explicit fmaf() in the same order, so the two sides agree bit-for-bit, while
accuracy against libm (double) is checked in tests/test_oracle_math.py.

Method: weighted least squares on dense Chebyshev nodes followed by a few
Remez-style exchange sweeps done in float64; coefficients are then rounded to
float32 and the float32/fma evaluation is error-checked on a dense grid.
"""
import numpy as np

def remez(f, basis, lo, hi, ncoef, iters=400, grid=8001):
    """Near-minimax fit by Lawson's iteratively re-weighted least squares:
    f(t) ~ sum_k c_k basis_k(t) on [lo, hi], minimising max |err|."""
    t = 0.5*(lo+hi) + 0.5*(hi-lo)*np.cos(np.pi*(np.arange(grid)+0.5)/grid)
    A = np.stack([basis(k, t) for k in range(ncoef)], 1)
    scale = np.max(np.abs(A), axis=0)          # column scaling for conditioning
    A = A / scale
    y = f(t)
    w = np.ones_like(t)
    best = (None, np.inf)
    for _ in range(iters):
        sw = np.sqrt(w)
        c, *_ = np.linalg.lstsq(A * sw[:, None], y * sw, rcond=None)
        err = np.abs(A @ c - y)
        m = err.max()
        if m < best[1]:
            best = (c / scale, m)
        w = w * (err + 1e-3 * m)
        w /= w.sum()
    return best

def f32(x): return np.float32(x)

def show(name, c):
    print(f"{name}:")
    for k, v in enumerate(c):
        print(f"  c{k} = {np.float32(v)!r:>18}   {float(np.float32(v)).hex()}")

# All fits minimise the ABSOLUTE error of the final function (unweighted), with the
# leading exactly-representable terms fixed (r, 1 - z/2, a, 1 + r).
PI4 = np.pi/4 * 1.01

# ---- sin on [0, pi/4]: sin(r) = r + sum_k s_k r^(3+2k)
cS, eS = remez(lambda r: np.sin(r) - r, lambda k, r: r**(3+2*k), 0.0, PI4, 3)
show("SIN: sin r = r + r*z*(s0 + z*(s1 + z*s2)), z=r*r", cS); print("  abs err", eS)

# ---- cos on [0, pi/4]: cos(r) = 1 - z/2 + sum_k c_k r^(4+2k)
cC, eC = remez(lambda r: np.cos(r) - 1 + r*r/2, lambda k, r: r**(4+2*k), 0.0, PI4, 3)
show("COS: cos r = 1 - z/2 + z*z*(c0 + z*(c1 + z*c2))", cC); print("  abs err", eC)

# ---- atan on [0,1]: atan(a) = a + sum_k p_k a^(3+2k)
for n in (7, 8, 9):
    cP, eP = remez(lambda a: np.arctan(a) - a, lambda k, a: a**(3+2*k), 0.0, 1.0, n)
    print("ATAN ncoef", n, "abs err", eP)
cP, eP = remez(lambda a: np.arctan(a) - a, lambda k, a: a**(3+2*k), 0.0, 1.0, 8)
show("ATAN: atan a = a + a*z*P(z), z=a*a, P degree 7", cP); print("  abs err", eP)

# ---- exp on [-ln2/2, ln2/2]: exp(r) = 1 + r + sum_k q_k r^(2+k)
L = np.log(2)/2 * 1.01
cQ, eQ = remez(lambda r: np.exp(r) - 1 - r, lambda k, r: r**(2+k), -L, L, 5)
show("EXP: exp r = 1 + r + r*r*(q0 + r*(q1 + r*(q2 + r*(q3 + r*q4))))", cQ); print("  abs err", eQ)
