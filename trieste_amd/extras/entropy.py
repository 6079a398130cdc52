"""Entropy-search acquisition on the engine's posterior (reference trieste/acquisition/function/entropy.py):
MinValueEntropySearch (50-163), min_value_entropy_search (166-214), GIBBON (236-419), GibbonAcquisition
(422-436), gibbon_quality_term (439-500), gibbon_repulsion_term (503-619).

MI355X-first: both tails are evaluated from the sweep's (mean, var) by a tail kernel (``TGP_ACQ_MES`` /
``TGP_ACQ_GIBBON``), so they keep the fused arg-max / top-k / analytic value-and-gradient entry points.  GIBBON's
repulsion term -- in the reference a block determinant built from ``covariance_between_points(x, pending)`` and
``predict_joint(pending)`` for every query batch -- is the predictive variance of the model *conditioned on the
pending points*: a clone of the engine with the pending rows appended to its factor (``tgp_clone_from`` +
``tgp_append_data``), swept at full speed next to the base model (``tgp_set_repulsion``).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..data import Dataset
from ..space import SearchSpace
from ..acquisition.function import _posterior_tail, _require_engine
from ..acquisition.interface import SingleModelAcquisitionBuilder, SingleModelGreedyAcquisitionBuilder
from ..acquisition.sampler import ExactThompsonSampler, ThompsonSampler


def _check_samples(samples) -> np.ndarray:
    s = np.asarray(samples, dtype=np.float64)
    if s.ndim != 2:
        raise ValueError(f"samples must have rank 2, got shape {s.shape}")
    if len(s) == 0:
        raise ValueError("samples must not be empty")
    return s


def _check_sampler_args(num_samples: int, grid_size: int, min_value_sampler, who: str):
    if num_samples <= 0:
        raise ValueError(f"num_samples must be positive, got {num_samples}")
    if grid_size <= 0:
        raise ValueError(f"grid_size must be positive, got {grid_size}")
    if min_value_sampler is None:
        return ExactThompsonSampler(sample_min_value=True)
    if not min_value_sampler.sample_min_value:
        raise ValueError(f"{who} requires a min_value_sampler that samples minimum values, however the passed "
                         f"sampler has sample_min_value=False.")
    return min_value_sampler


def _min_value_samples(sampler: ThompsonSampler, model, search_space: SearchSpace, dataset: Optional[Dataset],
                       num_samples: int, grid_size: int) -> np.ndarray:
    """Samples of y* over the data's query points + ``grid_size`` random points (entropy.py:132-139)."""
    if dataset is None or len(dataset) == 0:
        raise ValueError("Dataset must be populated.")
    query_points = np.concatenate([dataset.query_points, search_space.sample(grid_size)], axis=0)
    return np.asarray(sampler.sample(model, num_samples, query_points), dtype=np.float64)


class _entropy_tail(_posterior_tail):
    """An engine-backed tail that needs the min-value samples: they are handle state, installed before every
    call (S doubles), so several acquisition functions can share one engine."""

    def __init__(self, model, samples):
        self._samples = _check_samples(samples)
        super().__init__(model, 0.0)

    def update(self, samples) -> None:
        """New samples in place (entropy.py:188-192, 472-476)."""
        self._samples = _check_samples(samples)

    @property
    def samples(self) -> np.ndarray:
        return self._samples

    def _prepare(self) -> None:
        self._engine.set_min_value_samples(self._samples)
        self._engine.set_repulsion(None)


class min_value_entropy_search(_entropy_tail):
    r"""Max-value entropy search adapted to minimisation (Wang & Jegelka 2017; entropy.py:166-214): the
    information gained about the objective's minimum value :math:`y^*` by evaluating at x,
    mean over samples of :math:`-\gamma r / 2 - \log\Phi(-\gamma)`, :math:`\gamma = (y^* - \mu) / \sigma`,
    :math:`r = \phi(\gamma) / \Phi(-\gamma)`."""

    _acq = "mes"


class MinValueEntropySearch(SingleModelAcquisitionBuilder):
    """Builder of :class:`min_value_entropy_search` (entropy.py:50-163): ``num_samples`` samples of the minimum
    value over the data + ``grid_size`` random points, by an exact Thompson sampler unless another min-value
    sampler (Gumbel, trajectory-based) is given."""

    def __init__(self, search_space: SearchSpace, num_samples: int = 5, grid_size: int = 1000,
                 min_value_sampler: Optional[ThompsonSampler] = None):
        self._min_value_sampler = _check_sampler_args(num_samples, grid_size, min_value_sampler,
                                                      "Minvalue Entropy Search")
        self._search_space = search_space
        self._num_samples = num_samples
        self._grid_size = grid_size

    def __repr__(self) -> str:
        return (f"MinValueEntropySearch({self._search_space!r}, {self._num_samples!r}, {self._grid_size!r}, "
                f"{self._min_value_sampler!r})")

    def _draw(self, model, dataset):
        return _min_value_samples(self._min_value_sampler, model, self._search_space, dataset, self._num_samples,
                                  self._grid_size)

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return min_value_entropy_search(model, self._draw(model, dataset))

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if dataset is None or len(dataset) == 0:
            raise ValueError("Dataset must be populated.")
        if not isinstance(function, min_value_entropy_search):
            raise ValueError("function must be a min_value_entropy_search instance")
        function.update(self._draw(model, dataset))
        return function


# ---- GIBBON ----------------------------------------------------------------------------------------------
def _check_gibbon_model(model):
    if not (hasattr(model, "covariance_between_points") and hasattr(model, "get_observation_noise")):
        raise NotImplementedError(f"GIBBON only works with models that support covariance_between_points and "
                                  f"get_observation_noise; received {model!r}")


class gibbon_quality_term(_entropy_tail):
    r"""GIBBON's quality term (Moss et al. 2021; entropy.py:439-500): the information each batch element provides
    about :math:`y^*`, :math:`-\tfrac12` mean over samples of :math:`\log(1 + \rho^2 r (\gamma - r))` with
    :math:`\rho^2 = \sigma_f^2 / (\sigma_f^2 + \sigma_n^2)`."""

    _acq = "gibbon"

    def __init__(self, model, samples):
        if not hasattr(model, "get_observation_noise"):
            raise ValueError("GIBBON only currently supports homoscedastic models with a likelihood variance.")
        super().__init__(model, samples)


class gibbon_repulsion_term:
    r"""GIBBON's repulsion term (entropy.py:503-619): :math:`r = \tfrac12 (\log|V| - \log y_{var})`, |V| the
    determinant of the predictive covariance of the candidate's observation given the m pending points, times
    :math:`1/m^2` if ``rescaled_repulsion``.  That determinant ratio is the observation variance of the model
    conditioned on the pending points, so the term owns such a model (a clone of the base engine with the pending
    rows appended) and compares the two variances."""

    def __init__(self, model, pending_points, rescaled_repulsion: bool = True):
        pts = self._check_pending(pending_points)
        if not hasattr(model, "get_observation_noise"):
            raise ValueError("GIBBON only currently supports homoscedastic models with a likelihood variance.")
        if not hasattr(model, "covariance_between_points"):
            raise AttributeError("GIBBON only supports models with a covariance_between_points method.")
        _require_engine(model, "gibbon_repulsion_term")
        from ..models import FantasizedGaussianProcessRegression

        self._model = model
        self._rescaled_repulsion = rescaled_repulsion
        self._pending_points = pts
        # variances do not depend on the observations: condition on zeros
        self._conditioned = FantasizedGaussianProcessRegression(model, Dataset(pts, np.zeros((len(pts), 1))))

    @staticmethod
    def _check_pending(pending_points) -> np.ndarray:
        pts = np.asarray(pending_points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError(f"pending_points must be of shape [m, D], got {pts.shape}")
        if len(pts) == 0:
            raise ValueError("pending_points must not be empty")
        return pts

    def update(self, pending_points, lipschitz_constant=None, eta=None) -> None:
        """New pending points (entropy.py:570-577); rows added to the previous ones are appended to the factor."""
        pts = self._check_pending(pending_points)
        self._pending_points = pts
        self._conditioned.update_fantasized_data(Dataset(pts, np.zeros((len(pts), 1))))

    @property
    def conditioned_engine(self):
        return self._conditioned.engine

    @property
    def weight(self) -> float:
        return (1.0 / len(self._pending_points)) ** 2 if self._rescaled_repulsion else 1.0

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim < 2 or x.shape[-2] != 1:
            raise ValueError("This penalization function cannot be calculated for batches of points.")
        pts = x[..., 0, :]
        noise = float(self._model.get_observation_noise())
        _, var = self._model.engine.predict(pts)                 # two sweeps on the GPU ...
        _, var_c = self._conditioned.engine.predict(pts)
        rep = 0.5 * (np.log(np.asarray(var_c) + noise) - np.log(np.asarray(var) + noise))  # ... one log-ratio each
        return (self.weight * rep)[..., None]


class GibbonAcquisition:
    """quality(x) + repulsion(x) (entropy.py:422-436).  With this module's two terms on the same model the sum is
    formed on the device (``tgp_set_repulsion``) and the object exposes the fused ``argmax`` / ``top_k`` /
    ``value_and_gradient``; other callables are added from the values they return."""

    _FUSED_API = ("argmax", "top_k", "value_and_gradient", "_engine")

    def __init__(self, quality_term, diversity_term):
        self._quality_term = quality_term
        self._diversity_term = diversity_term

    def _fused(self) -> bool:
        q, r = self._quality_term, self._diversity_term
        return (isinstance(q, gibbon_quality_term) and isinstance(r, gibbon_repulsion_term)
                and r._model.engine is q._engine)

    def _install(self) -> None:
        q, r = self._quality_term, self._diversity_term
        q._engine.set_min_value_samples(q.samples)
        q._engine.set_repulsion(r.conditioned_engine, r.weight)

    def _run(self, method, *args, **kwargs):
        q = self._quality_term
        self._install()
        try:
            return method(q, *args, **kwargs)
        finally:
            q._engine.set_repulsion(None)

    def __call__(self, x):
        if self._fused():
            return self._run(_posterior_tail._evaluate, x)
        return np.asarray(self._diversity_term(x)) + np.asarray(self._quality_term(x))

    def __getattr__(self, name):  # the fused entry points exist only when the sum runs on the device
        if name in GibbonAcquisition._FUSED_API and self.__dict__.get("_quality_term") is not None and self._fused():
            if name == "_engine":
                return self._quality_term._engine
            raw = {"argmax": _posterior_tail._argmax, "top_k": _posterior_tail._top_k,
                   "value_and_gradient": _posterior_tail._value_and_gradient}[name]
            return lambda *args, **kwargs: self._run(raw, *args, **kwargs)
        raise AttributeError(name)


class GIBBON(SingleModelGreedyAcquisitionBuilder):
    """General-purpose Information-Based Bayesian OptimisatioN (Moss et al. 2021; entropy.py:236-419): a cheap
    approximation of the information a *batch* provides about the objective's minimum, built greedily -- the
    quality term of each element plus a repulsion term against the points already chosen."""

    def __init__(self, search_space: SearchSpace, num_samples: int = 5, grid_size: int = 1000,
                 min_value_sampler: Optional[ThompsonSampler] = None, rescaled_repulsion: bool = True):
        self._min_value_sampler = _check_sampler_args(num_samples, grid_size, min_value_sampler, "GIBBON")
        self._search_space = search_space
        self._num_samples = num_samples
        self._grid_size = grid_size
        self._rescaled_repulsion = rescaled_repulsion
        self._min_value_samples = None
        self._quality_term: Optional[gibbon_quality_term] = None
        self._diversity_term: Optional[gibbon_repulsion_term] = None
        self._gibbon_acquisition: Optional[GibbonAcquisition] = None

    def __repr__(self) -> str:
        return (f"GIBBON({self._search_space!r}, {self._num_samples!r}, {self._grid_size!r}, "
                f"{self._min_value_sampler!r}, {self._rescaled_repulsion!r})")

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None, pending_points=None):
        _check_gibbon_model(model)
        if dataset is None or len(dataset) == 0:
            raise ValueError("Dataset must be populated.")
        acq = self._update_quality_term(dataset, model)
        if pending_points is not None and len(pending_points) != 0:
            acq = self._update_repulsion_term(acq, dataset, model, pending_points)
        return acq

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None, pending_points=None,
                                    new_optimization_step: bool = True):
        if dataset is None or len(dataset) == 0:
            raise ValueError("Dataset must be populated.")
        if self._quality_term is None:
            raise ValueError("update_acquisition_function called before prepare_acquisition_function")
        if new_optimization_step:
            self._update_quality_term(dataset, model)
        if pending_points is None:
            return self._quality_term  # no repulsion term required if no pending_points
        return self._update_repulsion_term(function, dataset, model, pending_points)

    def _update_repulsion_term(self, function, dataset, model, pending_points):
        pts = np.asarray(pending_points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError(f"pending_points must be of shape [m, D], got {pts.shape}")
        if self._gibbon_acquisition is not None and isinstance(self._diversity_term, gibbon_repulsion_term):
            self._diversity_term.update(pts)  # same objects, new values
            return self._gibbon_acquisition
        self._diversity_term = gibbon_repulsion_term(model, pts, rescaled_repulsion=self._rescaled_repulsion)
        self._gibbon_acquisition = GibbonAcquisition(self._quality_term, self._diversity_term)
        return self._gibbon_acquisition

    def _update_quality_term(self, dataset: Dataset, model):
        self._min_value_samples = _min_value_samples(self._min_value_sampler, model, self._search_space, dataset,
                                                     self._num_samples, self._grid_size)
        if self._quality_term is not None:
            self._quality_term.update(self._min_value_samples)
        else:
            self._quality_term = gibbon_quality_term(model, self._min_value_samples)
        return self._quality_term
