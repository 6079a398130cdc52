"""Synthetic objectives on [0,1]^d used by BASELINE's configs (reference
trieste/objectives/single_objectives.py: Branin 83-139, Ackley 434-473, Hartmann-6 476-512) and
``mk_observer`` (objectives/utils.py:34-63).  Inputs to the hot path, not part of it."""
from __future__ import annotations

import math

import numpy as np

from .data import OBJECTIVE, Dataset
from .space import Box


def _branin_internals(x, scale, translate):
    x = np.asarray(x, dtype=np.float64)
    x0 = x[..., :1] * 15.0 - 5.0
    x1 = x[..., 1:] * 15.0
    b = 5.1 / (4 * math.pi ** 2)
    c = 5 / math.pi
    r, s, t = 6, 10, 1 / (8 * math.pi)
    return scale * ((x1 - b * x0 ** 2 + c * x0 - r) ** 2 + s * (1 - t) * np.cos(x0) + translate)


def branin(x):
    """Branin-Hoo over [0,1]^2, [..., 2] -> [..., 1]."""
    return _branin_internals(x, 1.0, 10.0)


def scaled_branin(x):
    """Branin-Hoo rescaled to zero mean, unit variance over [0,1]^2."""
    return _branin_internals(x, 1 / 51.95, -44.81)


BRANIN_MINIMIZERS = (np.array([[-math.pi, 12.275], [math.pi, 2.275], [9.42478, 2.475]]) + [5.0, 0.0]) / 15.0
BRANIN_MINIMUM = np.array([0.397887])
SCALED_BRANIN_MINIMUM = np.array([-1.047393])
BRANIN_SEARCH_SPACE = Box([0.0, 0.0], [1.0, 1.0])

_A = np.array([[10.0, 3.0, 17.0, 3.5, 1.7, 8.0], [0.05, 10.0, 17.0, 0.1, 8.0, 14.0],
               [3.0, 3.5, 1.7, 10.0, 17.0, 8.0], [17.0, 8.0, 0.05, 10.0, 0.1, 14.0]])
_P = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886], [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
               [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.6650], [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
_a = np.array([1.0, 1.2, 3.0, 3.2])
HARTMANN_6_MINIMIZER = np.array([[0.20169, 0.150011, 0.476874, 0.275332, 0.311652, 0.6573]])
HARTMANN_6_MINIMUM = np.array([-3.32237])


def hartmann_6(x):
    """Hartmann-6 over [0,1]^6, [..., 6] -> [..., 1]."""
    x = np.asarray(x, dtype=np.float64)
    inner = -np.sum(_A * (x[..., None, :] - _P) ** 2, axis=-1)
    return -np.sum(_a * np.exp(inner), axis=-1, keepdims=True)


def ackley(x):
    """Ackley over [0,1]^d (the reference's ackley_5 formula with 1/5 -> 1/d), [..., d] -> [..., 1]."""
    x = (np.asarray(x, dtype=np.float64) - 0.5) * (32.768 * 2.0)
    d = x.shape[-1]
    e1 = -0.2 * np.sqrt(np.sum(x ** 2, -1) / d)
    e2 = np.sum(np.cos(2.0 * math.pi * x), -1) / d
    return (-20.0 * np.exp(e1) - np.exp(e2) + 20.0 + math.e)[..., None]


def ackley_5(x):
    x = np.asarray(x)
    if x.shape[-1] != 5:
        raise ValueError(f"ackley_5 expects [..., 5], got {x.shape}")
    return ackley(x)


def synthetic_problem(objective, d: int, N: int, seed: int = 1234):
    """The seeded synthetic regression problem of the benchmarks: X ~ U[0,1]^{N x d}
    (numpy PCG64), Y = objective(X) standardised to zero mean / unit variance -> (X [N, d], Y [N])."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, d))
    Yraw = np.asarray(objective(X), dtype=np.float64).reshape(N)
    return X, (Yraw - Yraw.mean()) / Yraw.std()


def default_lengthscales(d: int) -> np.ndarray:
    """build_gpr's initial lengthscales on the unit cube: 0.2 * (upper - lower) * sqrt(d)
    (reference models/gpflow/builders.py:41, 413-423)."""
    return np.full(d, 0.2 * math.sqrt(d))


def mk_observer(objective, key=OBJECTIVE):
    """Observer returning {key: Dataset(x, objective(x))} (or a bare Dataset when key is None)."""
    if key is None:
        return lambda qp: Dataset(qp, objective(qp))
    return lambda qp: {key: Dataset(qp, objective(qp))}
