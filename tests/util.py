"""Shared helpers for the parity tests (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gp_goldens.json")

RTOL = 1e-5  # BASELINE.json north_star: "within 1e-5 relative fp64 tolerance"
EPS = np.finfo(np.float64).eps


def load_goldens():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


def cancellation_floor(N: int, variance: float, noise: float) -> float:
    """Absolute floor for variance-derived quantities.

    var = k** - |L^-1 k*|^2 cancels catastrophically near training inputs: two correct fp64
    evaluations differ by about eps * cond(L)^2-ish * variance.  cond(K + noise I) <= 1 + N*variance/noise,
    so we allow 64 * eps * variance * (N + N*variance/noise) ... capped to stay meaningful.
    The reference itself only asserts atol=1e-5 here (tests/unit/models/gpflow/test_models.py:363-365)
    and clips at 1e-12 (models/gpflow/interface.py:123)."""
    cond = 1.0 + N * variance / noise
    return min(64.0 * EPS * variance * cond, 1e-6 * variance)


def assert_close(actual, desired, rtol=RTOL, atol=0.0, what=""):
    actual = np.asarray(actual, dtype=np.float64)
    desired = np.asarray(desired, dtype=np.float64)
    assert actual.shape == desired.shape, f"{what}: shape {actual.shape} vs {desired.shape}"
    err = np.abs(actual - desired)
    tol = rtol * np.abs(desired) + atol
    bad = ~(err <= tol)
    if np.any(bad):
        i = np.argmax(err - tol)
        raise AssertionError(
            f"{what}: {bad.sum()} / {bad.size} elements differ; worst at flat index {i}: "
            f"actual={actual.flat[i]!r} desired={desired.flat[i]!r} err={err.flat[i]:.3e} "
            f"tol={tol.flat[i] if np.ndim(tol) else tol:.3e}")
