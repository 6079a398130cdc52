"""EXTRAS (out of scope per SURVEY.md section 2 rows 5 / 22; moved here from ``trieste_amd.acquisition.function`` in round 6):
the small builders that ride on the same engine-backed posterior -- NegativePredictiveMean, ProbabilityOfFeasibility,
MakePositive, MultipleOptimismNegativeLowerConfidenceBound, PredictiveVariance, ExpectedConstrainedImprovement
(reference acquisition/function/function.py:375-386, 507-589, 592-809, 1230-1420, acquisition/function/
active_learning.py:49-130).  Host-side compositions of ``predict`` / the fused tails; nothing in the hot-path package imports
them."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..acquisition.function import (NegativeLowerConfidenceBound, _eta_from, _is_torch, _posterior_tail, _require_engine,
                                    expected_improvement, probability_below_threshold)
from ..acquisition.interface import (AcquisitionFunctionBuilder, AcquisitionFunctionClass, SingleModelAcquisitionBuilder,
                                     SingleModelVectorizedAcquisitionBuilder)
from ..data import Dataset
from ..sampler import JITTER

# ---- small siblings on the same posterior ---------------------------------------------------------------
class NegativePredictiveMean(NegativeLowerConfidenceBound):
    """The negative of the predictive mean: -LCB with beta = 0 (function.py:375-386)."""

    def __init__(self):
        super().__init__(beta=0.0)

    def __repr__(self) -> str:
        return "NegativePredictiveMean()"


class ProbabilityOfFeasibility(SingleModelAcquisitionBuilder):
    r"""P(c(x) < threshold) of a constraint model (function.py:421-478); values below the threshold are feasible."""

    def __init__(self, threshold):
        if np.ndim(threshold) != 0:
            raise ValueError(f"threshold must be a scalar, got shape {np.shape(threshold)}")
        self._threshold = float(threshold)

    def __repr__(self) -> str:
        return f"ProbabilityOfFeasibility({self._threshold!r})"

    @property
    def threshold(self) -> float:
        return self._threshold

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return probability_below_threshold(model, self._threshold)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        return function  # no need to update anything


class _softplus_of(AcquisitionFunctionClass):
    r"""x -> log(1 + exp(f(x))) of an acquisition function f (function.py:1946-1950).  The transform is strictly
    increasing, so the fused arg-max / top-k of an engine-backed f are f's own (values transformed on the way out)
    and the gradient is sigmoid(f) f'."""

    def __init__(self, base):
        self._base = base

    @staticmethod
    def _softplus(v):
        return np.logaddexp(0.0, np.asarray(v, dtype=np.float64))

    def __call__(self, x):
        return self._softplus(self._base(x))

    def __getattr__(self, name):
        base = self.__dict__.get("_base")
        if base is None or name not in ("argmax", "top_k", "value_and_gradient", "_engine") or not hasattr(base, name):
            raise AttributeError(name)
        if name == "_engine":
            return base._engine
        if name == "argmax":
            def argmax(points, index_base: int = 0):
                v, i, x = base.argmax(points, index_base)
                return float(self._softplus(v)), i, x
            return argmax
        if name == "top_k":
            def top_k(points, k: int, index_base: int = 0):
                v, i = base.top_k(points, k, index_base)
                return self._softplus(v), i
            return top_k

        def value_and_gradient(points):
            v, g = base.value_and_gradient(points)
            v = np.asarray(v, dtype=np.float64)
            return self._softplus(v), np.asarray(g) * (1.0 / (1.0 + np.exp(-v)))[..., None]
        return value_and_gradient


class MakePositive(SingleModelAcquisitionBuilder):
    r"""Turns a builder into one whose functions are strictly positive, via :math:`x \mapsto \log(1 + \exp(x))`
    (function.py:1914-1990) -- what local penalization needs of its base function."""

    def __init__(self, base_acquisition_function_builder: SingleModelAcquisitionBuilder):
        self._base_builder = base_acquisition_function_builder
        self._base_function = None

    def __repr__(self) -> str:
        return f"MakePositive({self._base_builder!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        self._base_function = self._base_builder.prepare_acquisition_function(model, dataset)
        return _softplus_of(self._base_function)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        up_fn = self._base_builder.update_acquisition_function(self._base_function, model, dataset)
        if up_fn is self._base_function:
            return function
        self._base_function = up_fn
        return _softplus_of(up_fn)


class multiple_optimism_lower_confidence_bound(AcquisitionFunctionClass):
    r"""MOLCB (Torossian et al. 2020; function.py:1857-1911): column b of a batch is a -LCB with its own beta_b =
    5 D Phi^-1(1/2 + b / (2 (B + 1))), b = 1..B -- a gradient of exploration/exploitation trade-offs.
    x [..., B, D] -> [..., B]; the B columns are B sweeps of the engine's -LCB tail."""

    def __init__(self, model, search_space_dim: int):
        if search_space_dim <= 0:
            raise ValueError(f"search_space_dim must be positive, got {search_space_dim}")
        self._search_space_dim = search_space_dim
        self._model = model
        self._engine = _require_engine(model, type(self).__name__)
        self._betas = None

    def _betas_for(self, batch_size: int) -> np.ndarray:
        if batch_size <= 0:
            raise ValueError(f"batch size must be positive, got {batch_size}")
        if self._betas is None:
            from scipy.special import ndtri

            spread = 0.5 + 0.5 * np.arange(1, batch_size + 1, dtype=np.float64) / (batch_size + 1.0)
            self._betas = 5.0 * self._search_space_dim * ndtri(spread)
        elif len(self._betas) != batch_size:
            raise ValueError(f"{type(self).__name__} requires a fixed batch size. Got batch size {batch_size} but "
                             f"previous batch size was {len(self._betas)}.")
        return self._betas

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim < 2:
            raise ValueError(f"x must be [..., B, D], got shape {x.shape}")
        betas = self._betas_for(x.shape[-2])
        cols = [np.asarray(self._engine.acq_values("nlcb", float(b), x[..., j, :])) for j, b in enumerate(betas)]
        return np.stack(cols, axis=-1)

    def value_and_gradient(self, x):
        """x [R, B, D] -> (values [R, B], gradients [R, B, D]); column b differentiates function b.  Batch-size-one
        optimizers pass [P, D] points: the single column B = 1 (values [P], gradients [P, D])."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 2:
            val, grad = self.value_and_gradient(x[:, None, :])
            return val[:, 0], grad[:, 0, :]
        betas = self._betas_for(x.shape[-2])
        pairs = [self._engine.acq_value_grad("nlcb", float(b), x[..., j, :]) for j, b in enumerate(betas)]
        return (np.stack([np.asarray(p[0]) for p in pairs], axis=-1),
                np.stack([np.asarray(p[1]) for p in pairs], axis=-2))


class MultipleOptimismNegativeLowerConfidenceBound(SingleModelVectorizedAcquisitionBuilder):
    """Vectorized builder of :class:`multiple_optimism_lower_confidence_bound` (function.py:1808-1854): with
    ``EfficientGlobalOptimization(..., num_query_points=B)`` the B batch elements are optimised independently."""

    def __init__(self, search_space):
        self._search_space = search_space

    def __repr__(self) -> str:
        return f"MultipleOptimismNegativeLowerConfidenceBound({self._search_space!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return multiple_optimism_lower_confidence_bound(model, self._search_space.dimension)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if not isinstance(function, multiple_optimism_lower_confidence_bound):
            raise ValueError("function must be a multiple_optimism_lower_confidence_bound instance")
        return function  # nothing to update


class predictive_variance(AcquisitionFunctionClass):
    """x [..., B, D] -> det of the joint predictive covariance of the batch, exp(logdet(cov + jitter)) with the
    jitter added to every entry as the reference does (function/active_learning.py:86-110); B = 1: the predictive
    variance.  The covariance blocks come from the engine's joint kernel; the B x B determinant is host algebra."""

    def __init__(self, model, jitter: float):
        if not hasattr(model, "predict_joint"):
            raise NotImplementedError(f"PredictiveVariance only works with models that support predict_joint; "
                                      f"received {model!r}")
        self._model = model
        self._jitter = jitter

    def __call__(self, x):
        _, cov = self._model.predict_joint(x)
        sign, logdet = np.linalg.slogdet(np.asarray(cov, dtype=np.float64) + self._jitter)
        return np.where(sign > 0, np.exp(logdet), np.nan)


class PredictiveVariance(SingleModelAcquisitionBuilder):
    """Builder of :class:`predictive_variance` for active learning (function/active_learning.py:36-83)."""

    def __init__(self, jitter: float = JITTER):
        self._jitter = jitter

    def __repr__(self) -> str:
        return f"PredictiveVariance(jitter={self._jitter!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return predictive_variance(model, self._jitter)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        return function  # no need to update anything


# ---- constrained improvement: two models, two engines ----------------------------------------------------
class _product_of(AcquisitionFunctionClass):
    """x -> f(x) g(x) for two acquisition functions on DIFFERENT models (engines): each factor is evaluated by
    its own engine, the product and the product-rule gradient are formed on the returned arrays.  No fused
    arg-max exists across two engines; optimizers take their generic path."""

    def __init__(self, first, second):
        self._first, self._second = first, second

    def __call__(self, x):
        return np.asarray(self._first(x), dtype=np.float64) * np.asarray(self._second(x), dtype=np.float64)

    def __getattr__(self, name):
        first, second = self.__dict__.get("_first"), self.__dict__.get("_second")
        if name != "value_and_gradient" or first is None or not (hasattr(first, name) and hasattr(second, name)):
            raise AttributeError(name)

        def value_and_gradient(points):
            a, da = first.value_and_gradient(points)
            b, db = second.value_and_gradient(points)
            a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
            return a * b, np.asarray(da) * b[..., None] + np.asarray(db) * a[..., None]

        return value_and_gradient


class ExpectedConstrainedImprovement(AcquisitionFunctionBuilder):
    """Expected constrained improvement (Gardner et al. 2014; function.py:608-783): EI over the best *feasible*
    observed point times the constraint function (e.g. ``ProbabilityOfFeasibility(...).using(CONSTRAINT)``); with no
    feasible point yet, the constraint function alone."""

    def __init__(self, objective_tag, constraint_builder: AcquisitionFunctionBuilder,
                 min_feasibility_probability: float = 0.5, search_space=None):
        if np.ndim(min_feasibility_probability) != 0:
            raise ValueError("min_feasibility_probability must be a scalar")
        if not 0.0 <= float(min_feasibility_probability) <= 1.0:
            raise ValueError(f"min_feasibility_probability must be in [0, 1], got {min_feasibility_probability}")
        if search_space is not None and getattr(search_space, "has_constraints", False):
            raise NotImplementedError("explicitly constrained search spaces are outside the engine's path")
        self._objective_tag = objective_tag
        self._constraint_builder = constraint_builder
        self._search_space = search_space
        self._min_feasibility_probability = float(min_feasibility_probability)
        self._constraint_fn = None
        self._expected_improvement_fn = None
        self._constrained_improvement_fn = None

    def __repr__(self) -> str:
        return (f"ExpectedConstrainedImprovement({self._objective_tag!r}, {self._constraint_builder!r}, "
                f"{self._min_feasibility_probability!r}, {self._search_space!r})")

    def _feasible_mean(self, models, datasets):
        if datasets is None:
            raise ValueError("datasets must be provided")
        objective_model, objective_dataset = models[self._objective_tag], datasets[self._objective_tag]
        if len(objective_dataset) == 0:
            raise ValueError("Expected improvement is defined with respect to existing points in the objective data, "
                             "but the objective data is empty.")
        pof = np.asarray(self._constraint_fn(objective_dataset.query_points[:, None, :]))
        is_feasible = (pof >= self._min_feasibility_probability).reshape(-1)
        if not is_feasible.any():
            return objective_model, None
        feasible_mean, _ = objective_model.predict(objective_dataset.query_points[is_feasible])
        return objective_model, np.asarray(feasible_mean)

    def _update_expected_improvement_fn(self, objective_model, feasible_mean) -> None:
        eta = float(np.min(feasible_mean))
        if self._expected_improvement_fn is None:
            self._expected_improvement_fn = expected_improvement(objective_model, eta)
        else:
            self._expected_improvement_fn.update(eta)

    def prepare_acquisition_function(self, models, datasets=None):
        if datasets is None:
            raise ValueError("datasets must be provided")
        self._constraint_fn = self._constraint_builder.prepare_acquisition_function(models, datasets=datasets)
        objective_model, feasible_mean = self._feasible_mean(models, datasets)
        if feasible_mean is None:
            return self._constraint_fn
        self._update_expected_improvement_fn(objective_model, feasible_mean)
        self._constrained_improvement_fn = _product_of(self._expected_improvement_fn, self._constraint_fn)
        return self._constrained_improvement_fn

    def update_acquisition_function(self, function, models, datasets=None):
        if datasets is None:
            raise ValueError("datasets must be provided")
        if self._constraint_fn is None:
            raise ValueError("update_acquisition_function called before prepare_acquisition_function")
        self._constraint_fn = self._constraint_builder.update_acquisition_function(self._constraint_fn, models,
                                                                                   datasets=datasets)
        objective_model, feasible_mean = self._feasible_mean(models, datasets)
        if feasible_mean is None:
            return self._constraint_fn
        self._update_expected_improvement_fn(objective_model, feasible_mean)
        if self._constrained_improvement_fn is None or self._constrained_improvement_fn._second is not self._constraint_fn:
            self._constrained_improvement_fn = _product_of(self._expected_improvement_fn, self._constraint_fn)
        return self._constrained_improvement_fn
