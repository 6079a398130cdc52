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


# ---- the rest of the reference's single-objective suite (objectives/single_objectives.py:186-660) -------------
# Each is [..., d] -> [..., 1] in float64 over the search space of its test-problem record below; scalings are the
# reference's (several are rescaled to the unit cube and to zero mean / unit variance following Picheny et al. 2013).
def _as(x, d: int, name: str) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    if x.shape[-1] != d:
        raise ValueError(f"{name} expects [..., {d}], got {x.shape}")
    return x


def simple_quadratic(x):
    """-(x0 + x1)^2 over [0, 1]^2 (:186-194)."""
    return -np.sum(_as(x, 2, "simple_quadratic"), axis=-1, keepdims=True) ** 2


def gramacy_lee(x):
    """sin(10 pi x) / (2 x) + (x - 1)^4 over [0.5, 2.5] (Gramacy & Lee 2012; :208-217)."""
    x = _as(x, 1, "gramacy_lee")
    return np.sin(10.0 * math.pi * x) / (2.0 * x) + (x - 1.0) ** 4


def logarithmic_goldstein_price(x):
    """Log Goldstein-Price, standardised over [0, 1]^2 (Picheny et al. 2013; :232-248)."""
    u = 4.0 * _as(x, 2, "logarithmic_goldstein_price") - 2.0
    a, b = u[..., :1], u[..., 1:]
    first = 1.0 + (a + b + 1.0) ** 2 * (19.0 - 14.0 * a + 3.0 * a ** 2 - 14.0 * b + 6.0 * a * b + 3.0 * b ** 2)
    second = 30.0 + (2.0 * a - 3.0 * b) ** 2 * (18.0 - 32.0 * a + 12.0 * a ** 2 + 48.0 * b - 36.0 * a * b + 27.0 * b ** 2)
    return (np.log(first * second) - 8.693) / 2.427


_H3_A = np.array([[3.0, 10.0, 30.0], [0.1, 10.0, 35.0], [3.0, 10.0, 30.0], [0.1, 10.0, 35.0]])
_H3_P = np.array([[0.3689, 0.1170, 0.2673], [0.4699, 0.4387, 0.7470], [0.1091, 0.8732, 0.5547], [0.0381, 0.5743, 0.8828]])


def hartmann_3(x):
    """Hartmann-3 over [0, 1]^3 (:263-282)."""
    x = _as(x, 3, "hartmann_3")
    return -np.sum(_a * np.exp(-np.sum(_H3_A * (x[..., None, :] - _H3_P) ** 2, axis=-1)), axis=-1, keepdims=True)


_SHEKEL_BETA = np.array([0.1, 0.2, 0.2, 0.4, 0.4, 0.6, 0.3, 0.7, 0.5, 0.5])
_SHEKEL_C = np.array([[4.0, 1.0, 8.0, 6.0, 3.0, 2.0, 5.0, 8.0, 6.0, 7.0], [4.0, 1.0, 8.0, 6.0, 7.0, 9.0, 3.0, 1.0, 2.0, 3.6],
                      [4.0, 1.0, 8.0, 6.0, 3.0, 2.0, 5.0, 8.0, 6.0, 7.0], [4.0, 1.0, 8.0, 6.0, 7.0, 9.0, 3.0, 1.0, 2.0, 3.6]])


def shekel_4(x):
    """Shekel (10 wells) with [0, 10]^4 rescaled to [0, 1]^4 (:297-320)."""
    y = 10.0 * _as(x, 4, "shekel_4")
    wells = np.sum((y[..., :, None] - _SHEKEL_C) ** 2, axis=-2) + _SHEKEL_BETA  # [..., 10]
    return -np.sum(1.0 / wells, axis=-1, keepdims=True)


def levy(x, d: int):
    """Levy with [-10, 10]^d rescaled to [0, 1]^d (:336-359; the reference's inner term is sin(pi w + 1)^2)."""
    if d < 1:
        raise ValueError(f"d must be at least 1, got {d}")
    w = 1.0 + ((_as(x, d, "levy") * 20.0 - 10.0) - 1.0) / 4.0
    head = np.sin(math.pi * w[..., :1]) ** 2
    tail = (w[..., -1:] - 1.0) ** 2 * (1.0 + np.sin(2.0 * math.pi * w[..., -1:]) ** 2)
    wi = w[..., :-1]
    middle = np.sum((wi - 1.0) ** 2 * (1.0 + 10.0 * np.sin(math.pi * wi + 1.0) ** 2), axis=-1, keepdims=True)
    return head + middle + tail


def levy_8(x):
    """8-d Levy normalised to roughly the unit interval (:362-370)."""
    return levy(x, 8) / 450.0


def rosenbrock(x, d: int):
    """The reference's Rosenbrock variant with [-5, 10]^d rescaled to [0, 1]^d (:384-405):
    sum_i 100 (y_{i+1} - y_i)^2 + (1 - y_i)^2."""
    if d < 1:
        raise ValueError(f"d must be at least 1, got {d}")
    y = 15.0 * _as(x, d, "rosenbrock") - 5.0
    return np.sum(100.0 * (y[..., 1:] - y[..., :-1]) ** 2 + (1.0 - y[..., :-1]) ** 2, axis=-1, keepdims=True)


def rosenbrock_4(x):
    """4-d, standardised over [0, 1]^4 (:408-417)."""
    return (rosenbrock(x, 4) - 3.827e5) / 3.755e5


def michalewicz(x, d: int = 2, m: int = 10):
    """Michalewicz over [0, pi]^d with steepness m (:516-537)."""
    if d < 1:
        raise ValueError(f"d must be at least 1, got {d}")
    x = _as(x, d, "michalewicz")
    i = np.arange(1, d + 1, dtype=np.float64)
    return -np.sum(np.sin(x) * np.sin(i * x ** 2 / math.pi) ** (2 * m), axis=-1, keepdims=True)


def michalewicz_2(x):
    return michalewicz(x, 2)


def michalewicz_5(x):
    return michalewicz(x, 5)


def michalewicz_10(x):
    return michalewicz(x, 10)


def trid(x, d: int = 10):
    """Trid over [-d^2, d^2]^d (:616-635)."""
    if d < 2:
        raise ValueError(f"d must be at least 2, got {d}")
    x = _as(x, d, "trid")
    return np.sum((x - 1.0) ** 2, axis=-1, keepdims=True) - np.sum(x[..., 1:] * x[..., :-1], axis=-1, keepdims=True)


def trid_10(x):
    return trid(x, 10)


class SingleObjectiveTestProblem:
    """A synthetic test function with its search space, global minimizers and minimum (:39-75)."""

    def __init__(self, name: str, objective, search_space: Box, minimizers, minimum):
        self.name, self.objective, self.search_space = name, objective, search_space
        self.minimizers = np.asarray(minimizers, dtype=np.float64)
        self.minimum = np.asarray(minimum, dtype=np.float64)

    @property
    def dim(self) -> int:
        return self.search_space.dimension

    @property
    def bounds(self):
        return [self.search_space.lower, self.search_space.upper]

    def __repr__(self) -> str:
        return f"SingleObjectiveTestProblem({self.name!r}, dim={self.dim})"


def _unit(d: int) -> Box:
    return Box([0.0] * d, [1.0] * d)


Branin = SingleObjectiveTestProblem("Branin", branin, BRANIN_SEARCH_SPACE, BRANIN_MINIMIZERS, BRANIN_MINIMUM)
ScaledBranin = SingleObjectiveTestProblem("Scaled Branin", scaled_branin, BRANIN_SEARCH_SPACE, BRANIN_MINIMIZERS,
                                          SCALED_BRANIN_MINIMUM)
SimpleQuadratic = SingleObjectiveTestProblem("Simple Quadratic", simple_quadratic, _unit(2), [[1.0, 1.0]], [-4.0])
GramacyLee = SingleObjectiveTestProblem("Gramacy & Lee", gramacy_lee, Box([0.5], [2.5]), [[0.548562]], [-0.869011])
LogarithmicGoldsteinPrice = SingleObjectiveTestProblem("Logarithmic Goldstein-Price", logarithmic_goldstein_price,
                                                       _unit(2), [[0.5, 0.25]], [-3.12913])
Hartmann3 = SingleObjectiveTestProblem("Hartmann 3", hartmann_3, _unit(3), [[0.114614, 0.555649, 0.852547]], [-3.86278])
Shekel4 = SingleObjectiveTestProblem("Shekel 4", shekel_4, _unit(4), [[0.4, 0.4, 0.4, 0.4]], [-10.5363])
Levy8 = SingleObjectiveTestProblem("Levy 8", levy_8, _unit(8), [[11.0 / 20.0] * 8], [0.0])
Rosenbrock4 = SingleObjectiveTestProblem("Rosenbrock 4", rosenbrock_4, _unit(4), [[0.4] * 4], [-1.01917])
Ackley5 = SingleObjectiveTestProblem("Ackley 5", ackley_5, _unit(5), [[0.5] * 5], [0.0])
Hartmann6 = SingleObjectiveTestProblem("Hartmann 6", hartmann_6, _unit(6), HARTMANN_6_MINIMIZER, HARTMANN_6_MINIMUM)
_MICH = [2.202906, 1.570796, 1.284992, 1.923058, 1.720470, 1.570796, 1.454414, 1.756087, 1.655717, 1.570796]
Michalewicz2 = SingleObjectiveTestProblem("Michalewicz 2", michalewicz_2, Box([0.0] * 2, [math.pi] * 2), [_MICH[:2]],
                                          [-1.8013034])
Michalewicz5 = SingleObjectiveTestProblem("Michalewicz 5", michalewicz_5, Box([0.0] * 5, [math.pi] * 5), [_MICH[:5]],
                                          [-4.6876582])
Michalewicz10 = SingleObjectiveTestProblem("Michalewicz 10", michalewicz_10, Box([0.0] * 10, [math.pi] * 10), [_MICH],
                                           [-9.6601517])
Trid10 = SingleObjectiveTestProblem("Trid 10", trid_10, Box([-100.0] * 10, [100.0] * 10),
                                    [[i * (10 + 1 - i) for i in range(1, 11)]], [-10.0 * (10 + 4) * (10 - 1) / 6.0])

PROBLEMS = (Branin, ScaledBranin, SimpleQuadratic, GramacyLee, LogarithmicGoldsteinPrice, Hartmann3, Shekel4, Levy8,
            Rosenbrock4, Ackley5, Hartmann6, Michalewicz2, Michalewicz5, Michalewicz10, Trid10)
