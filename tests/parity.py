"""Shared parity metrics for the RHS tests.

Two criteria are used:

* ``rowwise_err``  - max |a-b| over a state row divided by max |b| over that row (a derivative
  can pass through zero at single nodes, so per-entry relative error is meaningless).  Used with
  the bound 1e-12 wherever the RHS is well conditioned (initial and synthetic states).
* ``backward_ok``  - near a steady state every row of the RHS is a small difference of large
  terms (convection vs. reaction, forward vs. reverse rate), so even two fp64 evaluations of the
  reference's own formula in different operation orders differ by ~1e-10 of the row maximum.
  There the test asks for backward stability instead: the result must lie within the change the
  oracle RHS itself shows when its input state is perturbed by 1e-13 relative.
"""
import numpy as np


def rowwise_err(a, b, V):
    a = np.asarray(a, float).reshape(V, -1)
    b = np.asarray(b, float).reshape(V, -1)
    den = np.max(np.abs(b), axis=1)
    den[den == 0] = 1.0
    return np.max(np.max(np.abs(a - b), axis=1)/den)


def backward_tol(fv, y, V, rel=1e-13, trials=4, seed=7):
    rng = np.random.default_rng(seed)
    y = np.asarray(y, float)
    f0 = fv(0.0, y).reshape(V, -1)
    tol = np.zeros(V)
    for _ in range(trials):
        yp = y*(1.0 + rel*rng.uniform(-1, 1, size=y.shape))
        tol = np.maximum(tol, np.max(np.abs(fv(0.0, yp).reshape(V, -1) - f0), axis=1))
    return np.maximum(tol, 1e-12*np.max(np.abs(f0), axis=1))


def backward_ok(f_test, f_ref, fv, y, V):
    """True if every row of f_test is within the oracle's own 1e-13-perturbation band of f_ref."""
    d = np.max(np.abs(np.asarray(f_test, float).reshape(V, -1) - np.asarray(f_ref, float).reshape(V, -1)), axis=1)
    return bool(np.all(d <= backward_tol(fv, y, V))), d
