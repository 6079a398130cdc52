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


def load_wide_qei_goldens():
    """mpmath goldens of batch Monte-Carlo EI at q = 9, 17, 33, 50 (oracle/make_goldens.py --wide-qei)."""
    with open(os.path.join(os.path.dirname(GOLDEN), "qei_wide_goldens.json")) as f:
        return json.load(f)["cases"]


def reparam_sample_atol(cov, floor, eps, jitter=1e-6):
    """Absolute tolerance of a reparametrised sample mean + (chol(cov + jitter I) eps): an error `floor` in the
    covariance entries moves the Cholesky factor by about floor / (2 sqrt(lambda_min)) per entry (first-order
    perturbation of the factorisation), times |eps| summed over a row."""
    cov = np.asarray(cov, dtype=np.float64)
    lam = min(float(np.linalg.eigvalsh(c + jitter * np.eye(c.shape[-1])).min()) for c in cov)
    q = cov.shape[-1]
    return floor + floor * q * float(np.abs(eps).max()) / (2.0 * np.sqrt(max(lam, jitter)))


def cancellation_floor(N: int, variance: float, noise: float) -> float:
    """Absolute floor for variance-derived quantities.

    var = k** - |L^-1 k*|^2 cancels catastrophically near training inputs: two correct fp64
    evaluations differ by about eps * cond(L)^2-ish * variance.  cond(K + noise I) <= 1 + N*variance/noise,
    so we allow 64 * eps * variance * (N + N*variance/noise) ... capped to stay meaningful.
    The reference itself only asserts atol=1e-5 here (tests/unit/models/gpflow/test_models.py:363-365)
    and clips at 1e-12 (models/gpflow/interface.py:123)."""
    cond = 1.0 + N * variance / noise
    return min(64.0 * EPS * variance * cond, 1e-6 * variance)


# ---- margins: observed worst error / tolerance of every parity comparison ------------------------------------------
# Every assert_close (and every record_margin of a hand-written comparison) notes how much of its tolerance the
# comparison used; tests/conftest.py writes the table at the end of the session (profiles/r03_parity_margins.txt is
# one such table from a GPU run).  A tolerance multiplier is justified by its row here, not by habit.
MARGINS = []  # (test id, what, worst error / tolerance, rtol, atol, elements)


def record_margin(what, err, tol, rtol=float("nan"), atol=float("nan")):
    err, tol = np.asarray(err, dtype=np.float64), np.asarray(tol, dtype=np.float64)
    if err.size == 0:
        return 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(err == 0.0, 0.0, err / tol)
    worst = float(np.nanmax(ratio)) if np.any(~np.isnan(ratio)) else float("nan")
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
    MARGINS.append((test, what, worst, float(np.max(rtol)), float(np.max(atol)), int(err.size)))
    return worst


def assert_close(actual, desired, rtol=RTOL, atol=0.0, what=""):
    actual = np.asarray(actual, dtype=np.float64)
    desired = np.asarray(desired, dtype=np.float64)
    assert actual.shape == desired.shape, f"{what}: shape {actual.shape} vs {desired.shape}"
    err = np.abs(actual - desired)
    tol = rtol * np.abs(desired) + atol
    record_margin(what, err, tol, rtol, atol)
    bad = ~(err <= tol)
    if np.any(bad):
        i = np.argmax(err - tol)
        raise AssertionError(
            f"{what}: {bad.sum()} / {bad.size} elements differ; worst at flat index {i}: "
            f"actual={actual.flat[i]!r} desired={desired.flat[i]!r} err={err.flat[i]:.3e} "
            f"tol={tol.flat[i] if np.ndim(tol) else tol:.3e}")


def i8x4_variance_bound(N: int, variance: float, w_abs_max: float) -> float:
    """Absolute error budget of the PLAIN split-precision sweep (TGP_PREC_I8X4 without the repair of TGP_PREC_AUTO;
    csrc/tgp_kernels_sweep_i8.inc) on the predictive variance, ON TOP of the parity tolerance: the digit pairs below
    2^-32 of S_i S' are dropped (S_i = (1 + 2^-7) max_k |W_ik| per row of W, S' = (1 + 2^-7) variance for K*), which
    leaves an error of rms 2^-32.8 S_i S' sqrt(i + 1) on c_i = (W k*)_i; with var = variance - |c|^2,
    |c| <= sqrt(variance):  |d var| ~ 2 |c| 2^-32.8 S' S_max sqrt(N) a priori (every row at the largest scale).  Twice
    that estimate is the budget; DESIGN.md section 4.5, tools/ozaki_tight.py."""
    tight = 1.0 + 2.0 ** -7
    s_max = tight * w_abs_max
    return 2.0 * (2.0 * np.sqrt(variance) * 2.0 ** -32.8 * (tight * variance) * s_max * np.sqrt(float(N)))
