"""Greedy batch acquisition on the engine's posterior (reference
trieste/acquisition/function/greedy_batch.py): LocalPenalization (54-247), PenalizedAcquisition (250-269),
local_penalizer / soft_local_penalizer / hard_local_penalizer (272-389), Fantasizer (415-585),
_generate_fantasized_data (588-609), _fantasized_model (630-773).

Both builders are driven by ``EfficientGlobalOptimization``'s greedy loop (rule.py:384-397): one acquisition
sweep per batch element.  MI355X-first:

* local penalization is a tail on the sweep: the engine multiplies the acquisition values by
  prod_p phi_p(x) on the device (``tgp_set_penalization``), so the penalized function keeps the fused arg-max /
  top-k / value-and-gradient entry points of the base function and a 10^6-candidate sweep never leaves HBM;
* a fantasized model is not a wrapper that re-derives the conditional posterior of every query batch from the
  base model (conditional_predict_f per call in the reference): it is a clone of the base engine with the
  fantasized rows *appended* to the cached factorisation (``tgp_clone_from`` + ``tgp_append_data``: O(k N^2)) --
  an exact GPR on (data + fantasized data), on which EI sweeps run at full speed.
"""
from __future__ import annotations

from typing import Callable, Mapping, Optional

import numpy as np

from ..data import OBJECTIVE, Dataset
from ..space import SearchSpace
from .function import ExpectedImprovement, _posterior_tail, _require_engine, expected_improvement
from .interface import (AcquisitionFunctionBuilder, GreedyAcquisitionFunctionBuilder, SingleModelAcquisitionBuilder,
                        SingleModelGreedyAcquisitionBuilder)


# ---- penalizers -----------------------------------------------------------------------------------
class local_penalizer:
    """Penalization around pending points with radius (mean(x') - eta) / L and scale sqrt(var(x')) / L
    (greedy_batch.py:272-312).  Evaluated by the engine (``tgp_penalization_values``); ``kind`` selects the
    device formula."""

    kind: str = ""

    def __init__(self, model, pending_points, lipschitz_constant, eta):
        self._model = model
        self._engine = _require_engine(model, type(self).__name__)
        self.update(pending_points, lipschitz_constant, eta)

    def update(self, pending_points, lipschitz_constant, eta) -> None:
        """New pending points / constants (greedy_batch.py:302-312)."""
        pts = np.asarray(pending_points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError(f"pending_points must be [P, D], got shape {pts.shape}")
        mean, var = self._model.predict(pts)
        lip = float(np.asarray(lipschitz_constant).reshape(()))
        eta = float(np.asarray(eta).reshape(()))
        self._pending_points = pts
        self._radius = (np.asarray(mean, dtype=np.float64).reshape(-1) - eta) / lip
        self._scale = np.sqrt(np.asarray(var, dtype=np.float64).reshape(-1)) / lip

    @property
    def parameters(self):
        """(pending points [P, D], radius [P], scale [P])."""
        return self._pending_points, self._radius, self._scale

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim < 2 or x.shape[-2] != 1:
            raise ValueError("This penalization function cannot be calculated for batches of points.")
        with self._engine.penalized(self.kind, *self.parameters):
            return np.asarray(self._engine.penalization_values(x[..., 0, :]))[..., None]


class soft_local_penalizer(local_penalizer):
    r"""phi(x, x') = Phi((|x - x'| - r) / s): the probability that x is outside the exclusion ball of the
    pending point (Gonzalez et al. 2016; greedy_batch.py:315-354)."""

    kind = "soft"


class hard_local_penalizer(local_penalizer):
    r"""phi(x, x') = ((|x - x'| / (r + s))^p + 1)^(1/p), p = -5 (Alvi et al. 2019; greedy_batch.py:357-389)."""

    kind = "hard"


class PenalizedAcquisition:
    """base(x) * penalization(x), in the reference computed as exp(log base + log penalization)
    (greedy_batch.py:250-269).  With an engine-backed base function and one of this module's penalizers on the
    same engine the product is formed on the device and the object exposes the base function's fused
    ``argmax`` / ``top_k`` / ``value_and_gradient``; any other pair of callables is combined from the values they
    return."""

    _FUSED_API = ("argmax", "top_k", "value_and_gradient", "_engine")

    def __init__(self, base_acquisition_function, penalization):
        self._base_acquisition_function = base_acquisition_function
        self._penalization = penalization

    def _fused(self) -> bool:
        base, pen = self._base_acquisition_function, self._penalization
        return (isinstance(base, _posterior_tail) and isinstance(pen, local_penalizer)
                and pen.kind in ("soft", "hard") and pen._engine is base._engine)

    def _scope(self):
        # a model with devices=[...] shards the base function's fused sweeps over its group: the penalization is
        # state of the handle, so it goes to EVERY member (member 0 alone would leave the other shards unpenalised
        # and the greedy batch would repeat its first point)
        pen, base = self._penalization, self._base_acquisition_function
        owner = base._group if getattr(base, "_group", None) is not None else base._engine
        return owner.penalized(pen.kind, *pen.parameters)

    def __call__(self, x):
        if self._fused():
            with self._scope():
                return self._base_acquisition_function(x)
        base = np.asarray(self._base_acquisition_function(x), dtype=np.float64)
        pen = np.asarray(self._penalization(x), dtype=np.float64)
        with np.errstate(divide="ignore"):
            return np.exp(np.log(base) + np.log(pen))

    def __getattr__(self, name):  # the fused entry points exist only when the product runs on the device
        if name in PenalizedAcquisition._FUSED_API and not name.startswith("__") and self.__dict__.get(
                "_base_acquisition_function") is not None and self._fused():
            if name == "_engine":
                return self._base_acquisition_function._engine
            target = getattr(self._base_acquisition_function, name)

            def call(*args, **kwargs):
                with self._scope():
                    return target(*args, **kwargs)

            return call
        raise AttributeError(name)


# ---- LocalPenalization ---------------------------------------------------------------------------------
class LocalPenalization(SingleModelGreedyAcquisitionBuilder):
    """Greedy batches by local penalization (greedy_batch.py:54-247): a strictly positive base acquisition
    function is down-weighted around the points already chosen; the size of the exclusion zones comes from
    a Lipschitz-constant estimate (the largest posterior-mean gradient norm over the data and
    ``num_samples`` random points), made once per optimisation step."""

    def __init__(self, search_space: SearchSpace, num_samples: int = 500,
                 penalizer: Optional[Callable] = None, base_acquisition_function_builder=None):
        if num_samples <= 0:
            raise ValueError(f"num_samples must be positive, got {num_samples}")
        self._search_space = search_space
        self._num_samples = num_samples
        self._lipschitz_penalizer = soft_local_penalizer if penalizer is None else penalizer
        self._base_builder = (ExpectedImprovement() if base_acquisition_function_builder is None
                              else base_acquisition_function_builder)
        self._lipschitz_constant = None
        self._eta = None
        self._base_acquisition_function = None
        self._penalization = None
        self._penalized_acquisition = None

    def __repr__(self) -> str:
        return (f"LocalPenalization({self._search_space!r}, {self._num_samples!r}, "
                f"{self._lipschitz_penalizer!r}, {self._base_builder!r})")

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None, pending_points=None):
        if dataset is None or len(dataset) == 0:
            raise ValueError("Dataset must be populated.")
        acq = self._update_base_acquisition_function(dataset, model)
        if pending_points is not None and len(pending_points) != 0:
            acq = self._update_penalization(acq, dataset, model, pending_points)
        return acq

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None, pending_points=None,
                                    new_optimization_step: bool = True):
        if dataset is None or len(dataset) == 0:
            raise ValueError("Dataset must be populated.")
        if self._base_acquisition_function is None:
            raise ValueError("update_acquisition_function called before prepare_acquisition_function")
        if new_optimization_step:
            self._update_base_acquisition_function(dataset, model)
        if pending_points is None or len(pending_points) == 0:
            return self._base_acquisition_function  # no penalization required
        return self._update_penalization(function, dataset, model, pending_points)

    def _update_penalization(self, function, dataset, model, pending_points):
        pts = np.asarray(pending_points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError(f"pending_points must be of shape [M, D], got {pts.shape}")
        if self._penalized_acquisition is not None and hasattr(self._penalization, "update"):
            self._penalization.update(pts, self._lipschitz_constant, self._eta)  # same objects, new values
            return self._penalized_acquisition
        self._penalization = self._lipschitz_penalizer(model, pts, self._lipschitz_constant, self._eta)
        self._penalized_acquisition = PenalizedAcquisition(self._base_acquisition_function, self._penalization)
        return self._penalized_acquisition

    @staticmethod
    def _get_lipschitz_estimate(model, sampled_points):
        """(max_i |d mean / dx (x_i)|, min_i mean(x_i)) (greedy_batch.py:206-217).  The mean and its gradient
        come from the engine's analytic value-and-gradient path: -LCB with beta = 0 is -mean."""
        eng = _require_engine(model, "LocalPenalization")
        neg_mean, neg_grad = eng.acq_value_grad("nlcb", 0.0, np.asarray(sampled_points, dtype=np.float64))
        grads_norm = np.linalg.norm(np.asarray(neg_grad), axis=1)
        return float(np.max(grads_norm)), float(np.min(-np.asarray(neg_mean)))

    def _update_base_acquisition_function(self, dataset: Dataset, model):
        samples = self._search_space.sample(self._num_samples)
        samples = np.concatenate([dataset.query_points, samples], axis=0)
        lipschitz_constant, eta = self._get_lipschitz_estimate(model, samples)
        if lipschitz_constant < 1e-5:  # numerical stability for 'flat' models
            lipschitz_constant = 10.0
        self._lipschitz_constant = lipschitz_constant
        self._eta = eta
        if self._base_acquisition_function is not None:
            self._base_acquisition_function = self._base_builder.update_acquisition_function(
                self._base_acquisition_function, model, dataset=dataset)
        elif isinstance(self._base_builder, ExpectedImprovement):  # reuse the eta estimate
            self._base_acquisition_function = expected_improvement(model, self._eta)
        else:
            self._base_acquisition_function = self._base_builder.prepare_acquisition_function(model, dataset=dataset)
        return self._base_acquisition_function


# ---- Fantasizer ----------------------------------------------------------------------------------------
def _generate_fantasized_data(fantasize_method: str, model, pending_points) -> Dataset:
    """"KB" (kriging believer): the model's mean at the pending points; "sample": one joint posterior sample
    (greedy_batch.py:588-609)."""
    if fantasize_method == "KB":
        fantasized_obs, _ = model.predict(pending_points)
    elif fantasize_method == "sample":
        fantasized_obs = model.sample(pending_points, num_samples=1)[0]
    else:
        raise NotImplementedError(f"fantasize_method must be KB or sample, received {fantasize_method!r}")
    return Dataset(pending_points, fantasized_obs)


def _fantasized_model(model, fantasized_data: Dataset):
    """A model conditioned on the base model's data AND ``fantasized_data`` (greedy_batch.py:630-773)."""
    from ..models import FantasizedGaussianProcessRegression

    return FantasizedGaussianProcessRegression(model, fantasized_data)


def _generate_fantasized_model(model, fantasized_data: Dataset):
    return _fantasized_model(model, fantasized_data)


def _supports_fantasizing(model) -> bool:
    eng = getattr(model, "engine", None)
    return (eng is not None and hasattr(eng, "clone") and hasattr(model, "predict_joint")
            and hasattr(model, "get_kernel") and hasattr(model, "get_observation_noise"))


class Fantasizer(GreedyAcquisitionFunctionBuilder):
    """Greedy batches with any non-batch acquisition function (greedy_batch.py:415-585): each chosen point gets
    a "fantasized" observation (KB: posterior mean, sample: a posterior draw) and the next element of the batch
    maximises the base acquisition function of the model conditioned on those."""

    def __init__(self, base_acquisition_function_builder=None, fantasize_method: str = "KB"):
        if fantasize_method not in ("KB", "sample"):
            raise ValueError(f"fantasize_method must be KB or sample, received {fantasize_method!r}")
        if base_acquisition_function_builder is None:
            base_acquisition_function_builder = ExpectedImprovement()
        if isinstance(base_acquisition_function_builder, SingleModelAcquisitionBuilder):
            base_acquisition_function_builder = base_acquisition_function_builder.using(OBJECTIVE)
        if not isinstance(base_acquisition_function_builder, AcquisitionFunctionBuilder):
            raise TypeError(f"unsupported base acquisition builder {base_acquisition_function_builder!r}")
        self._builder = base_acquisition_function_builder
        self._fantasize_method = fantasize_method
        self._base_acquisition_function = None
        self._fantasized_acquisition = None
        self._fantasized_models: Mapping = {}

    def __repr__(self) -> str:
        return f"Fantasizer({self._builder!r}, {self._fantasize_method!r})"

    def _update_base_acquisition_function(self, models, datasets):
        if self._base_acquisition_function is not None:
            self._base_acquisition_function = self._builder.update_acquisition_function(
                self._base_acquisition_function, models, datasets)
        else:
            self._base_acquisition_function = self._builder.prepare_acquisition_function(models, datasets)
        return self._base_acquisition_function

    def _update_fantasized_acquisition_function(self, models, datasets, pending_points):
        pts = np.asarray(pending_points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError(f"pending_points must be of shape [M, D], got {pts.shape}")
        fantasized_data = {tag: _generate_fantasized_data(self._fantasize_method, model, pts)
                           for tag, model in models.items()}
        if datasets is None:
            datasets = fantasized_data
        else:
            datasets = {tag: data + fantasized_data[tag] for tag, data in datasets.items()}
        if self._fantasized_acquisition is None:
            self._fantasized_models = {tag: _generate_fantasized_model(model, fantasized_data[tag])
                                       for tag, model in models.items()}
            self._fantasized_acquisition = self._builder.prepare_acquisition_function(self._fantasized_models,
                                                                                     datasets)
        else:
            for tag, model in self._fantasized_models.items():
                model.update_fantasized_data(fantasized_data[tag])
            self._fantasized_acquisition = self._builder.update_acquisition_function(
                self._fantasized_acquisition, self._fantasized_models, datasets)
        return self._fantasized_acquisition

    def prepare_acquisition_function(self, models, datasets=None, pending_points=None):
        for model in models.values():
            if not _supports_fantasizing(model):
                raise NotImplementedError(
                    f"Fantasizer only works with FastUpdateModel models that also support predict_joint, "
                    f"get_kernel and get_observation_noise (here: engine-backed GaussianProcessRegression); "
                    f"received {model!r}")
        if pending_points is None:
            return self._update_base_acquisition_function(models, datasets)
        return self._update_fantasized_acquisition_function(models, datasets, pending_points)

    def update_acquisition_function(self, function, models, datasets=None, pending_points=None,
                                    new_optimization_step: bool = True):
        if pending_points is None:
            return self._update_base_acquisition_function(models, datasets)
        return self._update_fantasized_acquisition_function(models, datasets, pending_points)
