"""Model wrapper of the hot path: ``GaussianProcessRegression`` over a GPU-resident engine.

Mirrors (names, argument meaning, error behaviour) the reference's
trieste/models/gpflow/models.py:69-526 (GaussianProcessRegression), interface.py:50-195
(GPflowPredictor) and builders.py:85-155 (build_gpr), with the GPflow objects it wraps replaced
by small parameter records (:class:`GPR`, :class:`Kernel`).  All arithmetic runs in libtgp (HIP);
there is no CPU path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from .data import Dataset
from .engine import GPEngine
from .space import SearchSpace

KERNEL_LENGTHSCALE = 0.2  # builders.py:41
SIGNAL_NOISE_RATIO_LIKELIHOOD = 10.0  # builders.py:78


# ---- parameter records standing in for gpflow.kernels.* / gpflow.models.GPR ----------------------
@dataclass
class Kernel:
    """Stationary kernel parameters (gpflow.kernels.Stationary): variance and (ARD) lengthscales."""

    variance: float = 1.0
    lengthscales: np.ndarray = field(default_factory=lambda: np.array(1.0))
    kind: str = "matern52"

    def __post_init__(self):
        self.variance = float(self.variance)
        self.lengthscales = np.asarray(self.lengthscales, dtype=np.float64)
        if not self.variance > 0 or np.any(self.lengthscales <= 0):
            raise ValueError("kernel variance and lengthscales must be positive")


def SquaredExponential(variance=1.0, lengthscales=1.0) -> Kernel:
    return Kernel(variance, lengthscales, "rbf")


RBF = SquaredExponential


def Matern12(variance=1.0, lengthscales=1.0) -> Kernel:
    return Kernel(variance, lengthscales, "matern12")


def Matern32(variance=1.0, lengthscales=1.0) -> Kernel:
    return Kernel(variance, lengthscales, "matern32")


def Matern52(variance=1.0, lengthscales=1.0) -> Kernel:
    return Kernel(variance, lengthscales, "matern52")


@dataclass
class Constant:
    """gpflow.mean_functions.Constant."""

    c: float = 0.0

    def __call__(self, x):
        x = np.asarray(x)
        return np.full(x.shape[:-1] + (1,), self.c)


@dataclass
class GPR:
    """Parameters of an exact GP regression model: the stand-in for ``gpflow.models.GPR``."""

    data: Tuple[np.ndarray, np.ndarray]
    kernel: Kernel
    mean_function: Constant = field(default_factory=Constant)
    likelihood_variance: float = 1.0

    def __post_init__(self):
        x = np.asarray(self.data[0], dtype=np.float64)
        y = np.asarray(self.data[1], dtype=np.float64)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError(f"data must be ([N, D], [N, 1]) arrays, got {x.shape}, {y.shape}")
        if y.shape[1] != 1:
            raise NotImplementedError("only single-output GPR is on the engine's path")
        self.data = (x, y)
        if not self.likelihood_variance > 0:
            raise ValueError("likelihood variance must be positive")


def build_gpr(data: Dataset, search_space: Optional[SearchSpace] = None, kernel_priors: bool = True,
              likelihood_variance: Optional[float] = None, trainable_likelihood: bool = False,
              kernel: Optional[Kernel] = None) -> GPR:
    """Sensible initial hyper-parameters (reference builders.py:85-155): Matern-5/2 with variance =
    empirical variance, lengthscales = 0.2 * (upper - lower) * sqrt(D) (1.0 on collapsed
    dimensions), constant mean = empirical mean, noise variance = variance / 10^2 unless given.
    ``kernel_priors`` / ``trainable_likelihood`` only matter for hyper-parameter fitting, which is
    outside this engine's path; they are accepted and ignored."""
    y = np.asarray(data.observations, dtype=np.float64)
    emp_mean, emp_var = float(np.mean(y)), float(np.var(y))
    if kernel is None:
        if search_space is None:
            raise ValueError("'build_gpr' function requires one of 'search_space' or 'kernel' arguments, but got neither")
        span = np.asarray(search_space.upper) - np.asarray(search_space.lower)
        ls = KERNEL_LENGTHSCALE * span * math.sqrt(search_space.dimension)
        ls = np.where(span == 0.0, 1.0, ls)
        kernel = Matern52(emp_var, ls)
    if likelihood_variance is None:
        noise = emp_var / SIGNAL_NOISE_RATIO_LIKELIHOOD ** 2
    else:
        if not likelihood_variance > 0:
            raise ValueError("likelihood_variance must be positive")
        noise = float(likelihood_variance)
    return GPR((data.query_points, data.observations), kernel, Constant(emp_mean), noise)


# ---- the model wrapper ------------------------------------------------------------------------
class GaussianProcessRegression:
    """Trainable-model protocol of the reference (models/interfaces.py:38-327) for an exact GPR,
    backed by :class:`~trieste_amd.engine.GPEngine`.

    ``update`` refreshes the posterior cache (K + s2 I -> L, L^-1, alpha) and does NOT train
    (reference interfaces.py:103-109); ``optimize`` (hyper-parameter fitting) is not part of this
    engine's path and keeps the current hyper-parameters.
    """

    def __init__(self, model: GPR, optimizer=None, num_kernel_samples: int = 10, num_rff_features: int = 1000,
                 use_decoupled_sampler: bool = True, device: int = 0):
        if num_kernel_samples < 0:
            raise ValueError(f"num_kernel_samples must be greater or equal to zero but got {num_kernel_samples}.")
        if num_rff_features <= 0:
            raise ValueError(f"num_rff_features must be greater than zero but got {num_rff_features}.")
        if not use_decoupled_sampler:
            raise NotImplementedError("only the decoupled trajectory sampler is on the engine's path")
        self._model = model
        self._num_kernel_samples = num_kernel_samples
        self._num_rff_features = num_rff_features
        self._use_decoupled_sampler = use_decoupled_sampler
        x, _ = model.data
        self._engine = GPEngine(x.shape[1], model.kernel.kind, device=device)
        self._push()

    def __repr__(self) -> str:
        return (f"GaussianProcessRegression({self._model!r}, {self._num_kernel_samples!r}, "
                f"{self._num_rff_features!r}, {self._use_decoupled_sampler!r})")

    # -- engine plumbing -------------------------------------------------------------------------
    def _push(self) -> None:
        m = self._model
        x, y = m.data
        self._engine.set_hyper(m.kernel.variance, np.broadcast_to(m.kernel.lengthscales, (x.shape[1],)),
                               m.likelihood_variance, m.mean_function.c)
        self._engine.set_data(x, y[:, 0])  # raises NotPositiveDefiniteError if the Cholesky fails

    @property
    def engine(self) -> GPEngine:
        return self._engine

    @property
    def model(self) -> GPR:
        return self._model

    # -- ProbabilisticModel -----------------------------------------------------------------------
    def predict(self, query_points):
        """[..., D] -> (mean [..., 1], var [..., 1]); var clipped to >= 1e-12 (interface.py:119-124)."""
        m, v = self._engine.predict(query_points)
        return m[..., None], v[..., None]

    def predict_joint(self, query_points):
        """[..., B, D] -> (mean [..., B, 1], cov [..., 1, B, B]) (interface.py:126-133)."""
        m, c = self._engine.predict_joint(query_points)
        return m[..., None], c[..., None, :, :]

    def predict_y(self, query_points):
        """Observation-space prediction: adds the Gaussian likelihood variance (models.py:167-169)."""
        m, v = self.predict(query_points)
        return m, v + self._model.likelihood_variance

    def sample(self, query_points, num_samples: int):
        """Exact joint samples (gpflow predict_f_samples): [..., N, D] -> [..., S, N, 1] = mean +
        chol(cov + jitter I) eps for N <= 64 query points (the engine's joint-posterior width);
        larger exact samples need the cross-covariance kernels SURVEY.md section 8f ranks as
        follow-up work.  The draws eps come from numpy's generator."""
        q = np.asarray(query_points, dtype=np.float64)
        if q.ndim < 2 or q.shape[-2] > 64:
            raise NotImplementedError("exact joint sampling is limited to [..., N <= 64, D] query points")
        eps = np.random.default_rng().normal(size=(q.shape[-2], int(num_samples)))
        return self._engine.reparam_samples(q, eps, 1e-6)[..., None]

    def log(self, dataset: Optional[Dataset] = None) -> None:
        """TensorBoard summaries are out of scope."""

    # -- TrainableProbabilisticModel ---------------------------------------------------------------
    def update(self, dataset: Dataset) -> None:
        """Assign new data and refresh the posterior cache (models.py:171-186)."""
        x, y = self._model.data
        qp, obs = np.asarray(dataset.query_points, np.float64), np.asarray(dataset.observations, np.float64)
        if qp.ndim != 2 or obs.ndim != 2 or qp.shape[0] != obs.shape[0]:
            raise ValueError(f"dataset must hold [N, D] query points and [N, L] observations, got {qp.shape}, {obs.shape}")
        if qp.shape[-1] != x.shape[-1]:
            raise ValueError(f"query points have dimension {qp.shape[-1]}, the model has {x.shape[-1]}")
        if obs.shape[-1] != y.shape[-1]:
            raise ValueError(f"observations have dimension {obs.shape[-1]}, the model has {y.shape[-1]}")
        self._model.data = (qp, obs)
        self._engine.set_data(qp, obs[:, 0])

    def optimize(self, dataset: Dataset):
        """Hyper-parameter fitting (models.py:256-321) is follow-up work: the
        hyper-parameters are kept and the posterior cache refreshed, as after a converged fit."""
        self.update(dataset)
        return None

    def set_hyperparameters(self, variance=None, lengthscales=None, likelihood_variance=None, mean=None) -> None:
        """Assign hyper-parameters (what a fit would do) and refresh the cache."""
        k = self._model.kernel
        if variance is not None:
            k.variance = float(variance)
        if lengthscales is not None:
            k.lengthscales = np.asarray(lengthscales, dtype=np.float64)
        if likelihood_variance is not None:
            self._model.likelihood_variance = float(likelihood_variance)
        if mean is not None:
            self._model.mean_function = Constant(float(mean))
        self._push()

    # -- Supports* protocols -------------------------------------------------------------------------
    def get_kernel(self) -> Kernel:
        return self._model.kernel

    def get_mean_function(self) -> Constant:
        return self._model.mean_function

    def get_observation_noise(self) -> float:
        return self._model.likelihood_variance

    def get_internal_data(self) -> Dataset:
        return Dataset(*self._model.data)

    # -- samplers -------------------------------------------------------------------------------------
    def reparam_sampler(self, num_samples: int):
        """interface.py:189-195."""
        from .sampler import BatchReparametrizationSampler

        return BatchReparametrizationSampler(num_samples, self)

    def trajectory_sampler(self):
        """models.py:323-345 (decoupled branch)."""
        from .sampler import DecoupledTrajectorySampler

        return DecoupledTrajectorySampler(self, self._num_rff_features)
