"""Acquisition functions of the hot path (reference trieste/acquisition/function/function.py):
ExpectedImprovement / expected_improvement (96-223), BatchMonteCarloExpectedImprovement /
batch_monte_carlo_expected_improvement (1074-1186), plus the sibling tails on the same posterior
ProbabilityOfImprovement (481-515, "probability_below_threshold"), NegativeLowerConfidenceBound
(328-418), AugmentedExpectedImprovement (226-325) and MonteCarloExpectedImprovement (786-920).  Values are computed by libtgp's fused kernels; the objects also expose the fused
device arg-max / top-k used by :mod:`trieste_amd.acquisition.optimizer`.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..data import Dataset
from ..sampler import JITTER
from .interface import (AcquisitionFunctionBuilder, AcquisitionFunctionClass, SingleModelAcquisitionBuilder,
                        SingleModelVectorizedAcquisitionBuilder)


def _require_engine(model, who: str):
    if not hasattr(model, "engine"):
        raise TypeError(f"{who} needs an engine-backed model (trieste_amd.models.GaussianProcessRegression); "
                        f"received {model!r}.  There is no CPU evaluation path.")
    return model.engine


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class _posterior_tail(AcquisitionFunctionClass):
    """Shared machinery of the analytic single-point acquisition functions: squeeze the batch
    axis (must be 1), call the fused kernel, restore [..., 1]."""

    _acq = "ei"

    def __init__(self, model, param: float):
        self._model = model
        self._engine = _require_engine(model, type(self).__name__)
        self._param = float(np.asarray(param).reshape(()))
        # a model built with devices=[...] drives one replica per GPU from this process: the fused sweeps of the
        # plain posterior tails then shard over the group (trieste_amd.group.GPEngineGroup)
        self._group = getattr(model, "group", None)

    def _shards(self) -> bool:
        """The group path serves the tails whose only state is (kind, param): a function that installs engine state
        of its own before a call (entropy tails, penalized wrappers act on member 0) stays on the single engine."""
        return self._group is not None and type(self)._prepare is _posterior_tail._prepare

    def _points(self, x):
        if not _is_torch(x):
            x = np.asarray(x, dtype=np.float64)
        if len(x.shape) < 2 or x.shape[-2] != 1:
            raise ValueError(f"This acquisition function only supports batch sizes of one, got input shape {tuple(x.shape)}")
        return x[..., 0, :]

    def _prepare(self) -> None:
        """Install per-function engine state before a call (the entropy tails' min-value samples)."""

    # the engine calls proper; wrappers that install their own engine state (GibbonAcquisition) use these
    def _evaluate(self, x):
        return self._engine.acq_values(self._acq, self._param, self._points(x))[..., None]

    def _value_and_gradient(self, points):
        return self._engine.acq_value_grad(self._acq, self._param, points)

    def _argmax(self, points, index_base: int = 0):
        return self._engine.acq_argmax(self._acq, self._param, points, index_base)

    def _top_k(self, points, k: int, index_base: int = 0):
        return self._engine.acq_topk(self._acq, self._param, points, k, index_base)

    def __call__(self, x):
        self._prepare()
        return self._evaluate(x)

    def value_and_gradient(self, points):
        """points [P, D] -> (values [P], d value / d point [P, D]): the pair
        tfp.math.value_and_gradient feeds L-BFGS-B with in the reference (optimizer.py:628-629)."""
        self._prepare()
        return self._value_and_gradient(points)

    # fused sweeps (no [M] values returned to the host)
    def argmax(self, points, index_base: int = 0):
        """points [M, D] -> (value, global index, point [D])."""
        self._prepare()
        if self._shards() and index_base == 0 and not _is_torch(points):
            self._group.set_candidates(points)
            return self._group.acq_argmax(self._acq, self._param)
        return self._argmax(points, index_base)

    def top_k(self, points, k: int, index_base: int = 0):
        self._prepare()
        if self._shards() and index_base == 0 and not _is_torch(points):
            self._group.set_candidates(points)
            return self._group.acq_topk(self._acq, self._param, k)
        return self._top_k(points, k, index_base)

    def argmax_pair(self, points, index_base: int = 0):
        """The fused arg-max with the winner left where the engine put it (a [2] device pair: value, global index
        bits) -- the sharded optimizers gather and merge such pairs without a host round trip per rank
        (``trieste_amd.distributed.all_gather_winners``).  Only for the tails whose state is (kind, param)."""
        if type(self)._prepare is not _posterior_tail._prepare:
            raise TypeError(f"{type(self).__name__} installs engine state per call: use argmax()")
        return self._engine.acq_argmax_pair(self._acq, self._param, points, index_base)

    def argmax_sampled(self, seed: int, num_samples: int, lower, upper):
        """Fused arg-max over ``num_samples`` uniform candidates of the box generated ON the device(s) (one logical
        Philox sample, sharded over the group when the model has one) -> (value, index, point [D])."""
        self._prepare()
        if self._shards():
            self._group.sample_candidates(seed, num_samples, lower, upper)
            return self._group.acq_argmax(self._acq, self._param)
        pts = self._engine.sample_box(seed, 0, num_samples, lower, upper)
        return self._argmax(pts, 0)


class expected_improvement(_posterior_tail):
    r"""x -> E[max(eta - f(x), 0)] (function.py:190-223)."""

    _acq = "ei"

    def __init__(self, model, eta):
        super().__init__(model, eta)

    def update(self, eta) -> None:
        """Update eta in place (no rebuild; the reference avoids retracing the same way)."""
        self._param = float(np.asarray(eta).reshape(()))

    @property
    def eta(self) -> float:
        return self._param


class probability_below_threshold(_posterior_tail):
    r"""x -> P(f(x) < threshold) (function.py:481-515)."""

    _acq = "pi"

    def update(self, threshold) -> None:
        self._param = float(np.asarray(threshold).reshape(()))


class negative_lower_confidence_bound(_posterior_tail):
    r"""x -> -(mean(x) - beta * sqrt(var(x))) (function.py:389-418)."""

    _acq = "nlcb"

    def __init__(self, model, beta: float = 1.96):
        if beta < 0:
            raise ValueError(f"Standard deviation scaling parameter beta must not be negative, got {beta}")
        super().__init__(model, beta)


class augmented_expected_improvement(_posterior_tail):
    r"""x -> EI(x) * (1 - sqrt(noise) / sqrt(noise + var(x))) (function.py:282-325); the observation
    noise is the model's likelihood variance, read by the kernel from the engine's state."""

    _acq = "aei"

    def __init__(self, model, eta):
        if not hasattr(model, "get_observation_noise"):
            raise NotImplementedError("AugmentedExpectedImprovement only works with models that support "
                                      f"get_observation_noise; received {model!r}")
        super().__init__(model, eta)

    def update(self, eta) -> None:
        self._param = float(np.asarray(eta).reshape(()))


def _eta_from(model, dataset: Optional[Dataset]) -> float:
    """min over the dataset's query points of the posterior MEAN (function.py:145-149)."""
    if dataset is None or len(dataset) == 0:
        raise ValueError("Dataset must be populated.")
    eng = _require_engine(model, "ExpectedImprovement")
    data = model.get_internal_data()
    if dataset.query_points.shape == data.query_points.shape and np.array_equal(dataset.query_points, data.query_points):
        return eng.eta()  # the model's own training inputs: fused on the device
    return float(np.min(eng.predict_mean(dataset.query_points)))


class ExpectedImprovement(SingleModelAcquisitionBuilder):
    """Builder for EI where the "best" value is the minimum posterior mean at the observed points
    (function.py:96-187).  Explicit search-space constraints are outside the engine's path."""

    def __init__(self, search_space=None):
        if search_space is not None and getattr(search_space, "has_constraints", False):
            raise NotImplementedError("constrained search spaces are outside the engine's path")
        self._search_space = search_space

    def __repr__(self) -> str:
        return f"ExpectedImprovement({self._search_space!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return expected_improvement(model, _eta_from(model, dataset))

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if not isinstance(function, expected_improvement):
            raise ValueError("function must be an expected_improvement instance")
        function.update(_eta_from(model, dataset))
        return function


class AugmentedExpectedImprovement(SingleModelAcquisitionBuilder):
    """Builder for augmented EI for noisy problems (function.py:226-279)."""

    def __repr__(self) -> str:
        return "AugmentedExpectedImprovement()"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        if not hasattr(model, "get_observation_noise"):
            raise NotImplementedError("AugmentedExpectedImprovement only works with models that support "
                                      f"get_observation_noise; received {model!r}")
        return augmented_expected_improvement(model, _eta_from(model, dataset))

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if not isinstance(function, augmented_expected_improvement):
            raise ValueError("function must be an augmented_expected_improvement instance")
        function.update(_eta_from(model, dataset))
        return function


class ProbabilityOfImprovement(SingleModelAcquisitionBuilder):
    """PI with threshold = min posterior mean at the observed points."""

    def __repr__(self) -> str:
        return "ProbabilityOfImprovement()"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return probability_below_threshold(model, _eta_from(model, dataset))

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        function.update(_eta_from(model, dataset))
        return function


class NegativeLowerConfidenceBound(SingleModelAcquisitionBuilder):
    def __init__(self, beta: float = 1.96):
        self._beta = beta

    def __repr__(self) -> str:
        return f"NegativeLowerConfidenceBound({self._beta!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return negative_lower_confidence_bound(model, self._beta)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        return function  # no state depends on the data


class batch_monte_carlo_expected_improvement(AcquisitionFunctionClass):
    """qEI = mean_S max(eta - min_q samples, 0) with reparametrised joint samples
    (function.py:1150-1186)."""

    def __init__(self, sample_size: int, model, eta, jitter: float):
        if not hasattr(model, "reparam_sampler"):
            raise ValueError("The batch Monte-Carlo expected improvement acquisition function only supports models "
                             f"that implement a reparam_sampler method; received {model!r}")
        self._sample_size = sample_size
        self._engine = _require_engine(model, type(self).__name__)
        self._sampler = model.reparam_sampler(sample_size)
        self._eta = float(np.asarray(eta).reshape(()))
        self._jitter = jitter

    def update(self, eta) -> None:
        """New eta and fresh draws (reference resets the reparam sampler)."""
        self._eta = float(np.asarray(eta).reshape(()))
        self._sampler.reset_sampler()

    def __call__(self, x):
        """x [..., B, D] -> [..., 1]."""
        if not _is_torch(x):
            x = np.asarray(x, dtype=np.float64)
        if len(x.shape) < 2:
            raise ValueError(f"x must be [..., B, D], got shape {tuple(x.shape)}")
        eps = self._sampler.eps(int(x.shape[-2]))
        if _is_torch(x) and x.is_cuda:
            import torch

            eps = torch.from_numpy(eps).to(x.device)
        return self._engine.qei(x, eps, self._eta, self._jitter)[..., None]


    def value_and_gradient(self, points):
        """points [P, q, D] -> (values [P], gradients [P, q, D]): the derivative TF autodiff takes through ``predict_joint``,
        ``tf.linalg.cholesky`` and the reparametrised samples when ``batchify_joint`` hands this function to the continuous
        optimizer (optimizer.py:628-629, 897-934; sampler.py:276-287), in reverse mode: the engine returns mean and covariance
        of the P groups as a skinny product (``joint_forward``) and, given the adjoints of both, the gradient w.r.t. the points
        (``joint_vjp``: K*, W K*, W^T (...) and the kernel derivatives over all N training rows); the q x q factorisations, the
        S-sample reduction and the Cholesky adjoint  A_bar = L^-T sym(Phi(L^T L_bar)) L^-1  in between are host arithmetic."""
        x = np.ascontiguousarray(points.cpu().numpy() if _is_torch(points) else points, dtype=np.float64)
        if x.ndim != 3:
            raise ValueError(f"points must be [P, q, D], got shape {x.shape}")
        P, q, _ = x.shape
        eps = np.asarray(self._sampler.eps(q), dtype=np.float64)                      # [q, S]
        S = eps.shape[1]
        epsT = np.ascontiguousarray(eps.T)
        values, grads = np.empty(P), np.empty(x.shape)
        chunk = max(1, getattr(self._engine, "JOINT_SMALL_POINTS", 2048) // q)
        eye = np.eye(q)
        on_device = hasattr(self._engine, "qei_value_grad") and self._engine.qei_value_grad_fits(q, S)
        for g0 in range(0, P, chunk):
            xs = x[g0:g0 + chunk]
            if on_device:   # the q x q arithmetic below on the device as well (one wave per group): one call per chunk
                v, g_ = self._engine.qei_value_grad(xs, eps, self._eta, self._jitter)
                values[g0:g0 + chunk], grads[g0:g0 + chunk] = np.asarray(v), np.asarray(g_)
                continue
            mean, cov = (np.asarray(a) for a in self._engine.joint_forward(xs))         # [g, q], [g, q, q]
            g = mean.shape[0]
            clipped = np.diagonal(cov, axis1=-2, axis2=-1) <= 1e-12                     # (clip_by_value: zero gradient)
            L = np.linalg.cholesky(cov + self._jitter * eye)
            smp = L @ eps                                                               # [g, q, S]: the sample index innermost
            smp += mean[:, :, None]
            low = smp[:, 0, :].copy()                                                   # running minimum over the batch points,
            jmin = np.zeros(low.shape, dtype=np.int64)                                  # its FIRST index on ties (tf.reduce_min's
            for i in range(1, q):                                                       # gradient goes to ... any one: measure zero)
                better = smp[:, i, :] < low
                jmin[better] = i
                np.minimum(low, smp[:, i, :], out=low)
            imp = self._eta - low
            active = imp > 0.0
            values[g0:g0 + chunk] = np.where(active, imp, 0.0).mean(axis=1)
            jmin[~active] = -1
            # adjoints: d value / d sample[g, i, s] = -1/S where i is the arg-min of an improving sample, else 0
            gmean, Lbar = np.empty((g, q)), np.zeros((g, q, q))
            for i in range(q):
                w = (jmin == i).astype(np.float64)                                      # [g, S]
                gmean[:, i] = w.sum(axis=-1)
                Lbar[:, i, :i + 1] = w @ epsT[:, :i + 1]                                # lower triangle of sum_s d sample eps^T
            gmean *= -1.0 / S
            Lbar *= -1.0 / S
            Lt = np.swapaxes(L, -1, -2)
            Q = np.tril(Lt @ Lbar)
            Q[:, np.arange(q), np.arange(q)] *= 0.5
            R = 0.5 * (Q + np.swapaxes(Q, -1, -2))
            Y = np.linalg.solve(Lt, R)                                                  # L^-T R
            gcov = np.swapaxes(np.linalg.solve(Lt, np.swapaxes(Y, -1, -2)), -1, -2)     # L^-T R L^-1
            gcov[:, np.arange(q), np.arange(q)] = np.where(clipped, 0.0, np.diagonal(gcov, axis1=-2, axis2=-1))
            grads[g0:g0 + chunk] = np.asarray(self._engine.joint_vjp(xs, gmean, np.ascontiguousarray(gcov)))
        return values, grads


class BatchMonteCarloExpectedImprovement(SingleModelAcquisitionBuilder):
    """Builder for qEI (function.py:1074-1147)."""

    def __init__(self, sample_size: int, *, jitter: float = JITTER):
        if sample_size <= 0:
            raise ValueError(f"sample_size must be positive, got {sample_size}")
        if jitter < 0:
            raise ValueError(f"jitter must be non-negative, got {jitter}")
        self._sample_size = sample_size
        self._jitter = jitter

    def __repr__(self) -> str:
        return f"BatchMonteCarloExpectedImprovement({self._sample_size!r}, jitter={self._jitter!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return batch_monte_carlo_expected_improvement(self._sample_size, model, _eta_from(model, dataset), self._jitter)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if not isinstance(function, batch_monte_carlo_expected_improvement):
            raise ValueError("function must be a batch_monte_carlo_expected_improvement instance")
        function.update(_eta_from(model, dataset))
        return function


class monte_carlo_expected_improvement(AcquisitionFunctionClass):
    r"""x -> mean_S max(eta - f_s(x), 0) from reparametrised samples (function.py:883-920).  For the
    exact GPR ``model.reparam_sampler`` is the batch sampler at batch size one, so this is the qEI
    kernel with q = 1."""

    def __init__(self, sampler, eta, jitter: float = JITTER):
        self._sampler = sampler
        self._engine = _require_engine(sampler._model, type(self).__name__)
        self._eta = float(np.asarray(eta).reshape(()))
        self._jitter = jitter

    def update(self, eta) -> None:
        self._eta = float(np.asarray(eta).reshape(()))

    def __call__(self, at):
        if not _is_torch(at):
            at = np.asarray(at, dtype=np.float64)
        if len(at.shape) < 2 or at.shape[-2] != 1:
            raise ValueError(f"This acquisition function only supports batch sizes of one, got input shape {tuple(at.shape)}")
        eps = self._sampler.eps(1)
        if _is_torch(at) and at.is_cuda:
            import torch

            eps = torch.from_numpy(eps).to(at.device)
        return self._engine.qei(at, eps, self._eta, self._jitter)[..., None]

    def value_and_gradient(self, points):
        """points [P, D] -> (values [P], gradients [P, D]): the derivative TF autodiff takes through the
        reparametrised samples mu(x) + sqrt(var(x) + jitter) eps_s (optimizer.py:628-629), in closed form:
        mean_s 1[eta > sample_s] (-dmu - eps_s ds), ds = dvar / (2 sqrt(var + jitter)).  Mean, sd and their
        gradients come from the engine's analytic path (-LCB at beta = 0 and 1); the S-sample reduction over the
        few hundred L-BFGS-B iterates is host arithmetic."""
        pts = np.asarray(points, dtype=np.float64)
        eps = np.asarray(self._sampler.eps(1), dtype=np.float64).reshape(-1)          # [S]
        neg_mu, neg_dmu = (np.asarray(a) for a in self._engine.acq_value_grad("nlcb", 0.0, pts))
        lcb1, dlcb1 = (np.asarray(a) for a in self._engine.acq_value_grad("nlcb", 1.0, pts))
        mu, dmu = -neg_mu, -neg_dmu
        sd, dsd = lcb1 - neg_mu, dlcb1 - neg_dmu                                        # sqrt(var), its gradient
        s = np.sqrt(sd * sd + self._jitter)
        ds = (sd / s)[:, None] * dsd                                                    # d sqrt(var + jitter)
        improvement = self._eta - (mu[:, None] + s[:, None] * eps[None, :])             # [P, S]
        active = improvement > 0.0
        value = np.mean(np.where(active, improvement, 0.0), axis=1)
        weight = active.mean(axis=1)                                                    # mean_s 1[...]
        weight_eps = (active * eps[None, :]).mean(axis=1)                               # mean_s 1[...] eps_s
        return value, -weight[:, None] * dmu - weight_eps[:, None] * ds


class MonteCarloExpectedImprovement(SingleModelAcquisitionBuilder):
    """Builder for Monte-Carlo EI; eta = min over the data of the SAMPLE mean (function.py:786-880)."""

    def __init__(self, sample_size: int, *, jitter: float = JITTER):
        if sample_size <= 0:
            raise ValueError(f"sample_size must be positive, got {sample_size}")
        if jitter < 0:
            raise ValueError(f"jitter must be non-negative, got {jitter}")
        self._sample_size = sample_size
        self._jitter = jitter

    def __repr__(self) -> str:
        return f"MonteCarloExpectedImprovement({self._sample_size!r}, jitter={self._jitter!r})"

    def _eta(self, sampler, dataset) -> float:
        if dataset is None or len(dataset) == 0:
            raise ValueError("Dataset must be populated.")
        samples = sampler.sample(np.asarray(dataset.query_points)[..., None, :], jitter=self._jitter)  # [N, S, 1, 1]
        return float(np.min(np.mean(np.asarray(samples), axis=-3)))

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        if not hasattr(model, "reparam_sampler"):
            raise ValueError("MonteCarloExpectedImprovement only supports models with a reparam_sampler method; "
                             f"received {model!r}")
        sampler = model.reparam_sampler(self._sample_size)
        return monte_carlo_expected_improvement(sampler, self._eta(sampler, dataset), self._jitter)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if not isinstance(function, monte_carlo_expected_improvement):
            raise ValueError("function must be a monte_carlo_expected_improvement instance")
        function._sampler.reset_sampler()
        function.update(self._eta(function._sampler, dataset))
        return function
