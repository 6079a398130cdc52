"""Model samplers of the hot path (reference trieste/models/gpflow/sampler.py):
``BatchReparametrizationSampler`` (167-287) and the decoupled trajectory sampler
(``DecoupledTrajectorySampler`` 594-738, ``feature_decomposition_trajectory`` 858-953).

The random draws (eps; the RFF basis W, b; the weights w, xi) are made HERE with numpy's PCG64 and
handed to the engine -- TF's Philox streams are not reproducible outside TF, so parity is defined
with the draws as explicit inputs.  All arithmetic on them runs on the GPU.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from .rng import make_rng

JITTER = 1e-6  # reference utils/misc.py:180-183


def qmc_normal_samples(num_samples: int, n_sample_dim: int, skip: int = 0) -> np.ndarray:
    """``num_samples`` Sobol points of dimension ``n_sample_dim`` (the first ``skip`` skipped) pushed through
    the normal quantile function -> [num_samples, n_sample_dim] (reference sampler.py:53-79:
    ``tf.math.sobol_sample`` + ``Normal.quantile``).  scipy's unscrambled Sobol generator is the same
    direction-number sequence; TF's stream starts after the all-zero point, hence the ``+ 1``."""
    if num_samples == 0 or n_sample_dim == 0:
        return np.zeros((num_samples, n_sample_dim))
    from scipy.special import ndtri
    from scipy.stats import qmc

    gen = qmc.Sobol(d=n_sample_dim, scramble=False)
    gen.fast_forward(int(skip) + 1)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # scipy warns when num_samples is not a power of two
        pts = gen.random(num_samples)
    return ndtri(pts)


class _QmcSkip:
    """The class-wide Sobol skip counter shared by the reparametrization samplers (reference
    ``IndependentReparametrizationSampler.skip``, sampler.py:95-96, incremented per (re)draw)."""

    skip = 0


class BatchReparametrizationSampler:
    r"""x -> mu(x) + L(x) eps with eps ~ N(0, 1) fixed until :meth:`reset_sampler`, so samples form
    a continuous surface (reference sampler.py:167-287)."""

    def __init__(self, sample_size: int, model, qmc: bool = False, qmc_skip: bool = True,
                 seed: Optional[int] = None):
        if sample_size <= 0:
            raise ValueError(f"sample_size must be positive, got {sample_size}")
        if not hasattr(model, "predict_joint"):
            raise NotImplementedError(
                f"BatchReparametrizationSampler only works with models that support predict_joint; received {model!r}")
        self._qmc, self._qmc_skip = qmc, qmc_skip
        self._sample_size = sample_size
        self._model = model
        self._rng = make_rng(seed)
        self._eps: Optional[np.ndarray] = None  # [B, S]
        self._initialized = False

    def __repr__(self) -> str:
        return f"BatchReparametrizationSampler({self._sample_size!r}, {self._model!r})"

    @property
    def sample_size(self) -> int:
        return self._sample_size

    def reset_sampler(self) -> None:
        """Resample eps on the next call (reference ReparametrizationSampler.reset_sampler)."""
        self._initialized = False

    def eps(self, batch_size: int) -> np.ndarray:
        """The fixed draws [B, S] for this batch size (drawn on first use / after a reset)."""
        if batch_size <= 0:
            raise ValueError(f"batch size must be positive, got {batch_size}")
        if not self._initialized or self._eps is None:
            if self._qmc:  # Sobol points [S, B] -> [B, S] (sampler.py:241-256)
                skip = _QmcSkip.skip if self._qmc_skip else 0
                if self._qmc_skip:
                    _QmcSkip.skip += self._sample_size
                self._eps = np.ascontiguousarray(qmc_normal_samples(self._sample_size, batch_size, skip).T)
            else:
                self._eps = self._rng.standard_normal((batch_size, self._sample_size))
            self._initialized = True
        elif self._eps.shape[0] != batch_size:
            raise ValueError(f"{type(self).__name__} requires a fixed batch size. Got batch size {batch_size} "
                             f"but previous batch size was {self._eps.shape[0]}.")
        return self._eps

    def sample(self, at, *, jitter: float = JITTER):
        """at [..., B, D] -> samples [..., S, B, 1]; identical across calls for the same ``at``."""
        at = np.asarray(at, dtype=np.float64)
        if at.ndim < 2:
            raise ValueError(f"at must have rank >= 2, got shape {at.shape}")
        if jitter < 0:
            raise ValueError(f"jitter must be non-negative, got {jitter}")
        eps = self.eps(at.shape[-2])
        return self._model.engine.reparam_samples(at, eps, jitter)[..., None]


class IndependentReparametrizationSampler:
    r"""x -> mu(x) + sqrt(var(x) + jitter) eps with one fixed eps [S] (reference sampler.py:82-164):
    samples of the MARGINAL posteriors, continuous in x.  On the engine this is the reparametrised
    sample kernel at batch size one (the Cholesky factor of the 1 x 1 matrix var + jitter)."""

    def __init__(self, sample_size: int, model, qmc: bool = False, qmc_skip: bool = True,
                 seed: Optional[int] = None):
        if sample_size <= 0:
            raise ValueError(f"sample_size must be positive, got {sample_size}")
        self._qmc, self._qmc_skip = qmc, qmc_skip
        self._sample_size = sample_size
        self._model = model
        self._rng = make_rng(seed)
        self._eps: Optional[np.ndarray] = None  # [1, S]
        self._initialized = False

    def __repr__(self) -> str:
        return f"IndependentReparametrizationSampler({self._sample_size!r}, {self._model!r})"

    def reset_sampler(self) -> None:
        self._initialized = False

    def sample(self, at, *, jitter: float = JITTER):
        """at [..., 1, D] -> samples [..., S, 1, 1] = mean + sqrt(var + jitter) eps (sampler.py:137-164)."""
        at = np.asarray(at, dtype=np.float64)
        if at.ndim < 2 or at.shape[-2] != 1:
            raise ValueError(f"at must be [..., 1, D], got shape {at.shape}")
        if jitter < 0:
            raise ValueError(f"jitter must be non-negative, got {jitter}")
        if not self._initialized or self._eps is None:
            if self._qmc:
                skip = _QmcSkip.skip if self._qmc_skip else 0
                if self._qmc_skip:
                    _QmcSkip.skip += self._sample_size
                self._eps = np.ascontiguousarray(qmc_normal_samples(self._sample_size, 1, skip).T)
            else:
                self._eps = self._rng.standard_normal((1, self._sample_size))
            self._initialized = True
        return self._model.engine.reparam_samples(at, self._eps, jitter)[..., None]


# ---- decoupled trajectories -------------------------------------------------------------------
_MATERN_DOF = {"matern12": 1, "matern32": 3, "matern52": 5}


def sample_rff_basis(kind: str, num_features: int, dim: int, rng: np.random.Generator):
    """Spectral draws of gpflux ``RandomFourierFeaturesCosine``: W [F, D] rows ~ N(0, I) for the
    squared exponential, multivariate Student-t with 2 nu degrees of freedom for Matern-nu
    (N(0, I) * sqrt(dof / chi2_dof), one chi2 per feature); b [F] ~ U(0, 2 pi)."""
    W = rng.standard_normal((num_features, dim))
    if kind != "rbf":
        dof = _MATERN_DOF[kind]
        W = W * np.sqrt(dof / rng.chisquare(dof, size=(num_features, 1)))
    b = rng.uniform(0.0, 2.0 * math.pi, size=num_features)
    return W, b


class decoupled_trajectory:
    """A batch of B approximate posterior draws f_b(x) = sum_f phi_f(x) w_fb + sum_j k(x, X_j) v_jb + m(x)
    (reference ``feature_decomposition_trajectory`` sampler.py:858-953).  The batch size is fixed by
    the first call; weights are (re)sampled lazily."""

    def __init__(self, sampler: "DecoupledTrajectorySampler"):
        self._sampler = sampler
        self._traj = None  # engine Trajectory
        self._batch_size = 0

    def _ensure(self, B: int) -> None:
        if self._traj is None:
            self._batch_size = B
            self.resample()
        elif B != self._batch_size:
            raise ValueError(f"This trajectory only supports batch sizes of {self._batch_size}. If you wish to change "
                             "the batch size you must get a new trajectory by calling the get_trajectory method of "
                             "the trajectory sampler.")

    def resample(self) -> None:
        """New weights (w, xi) for the current basis; v is solved on the GPU from the cached factor."""
        s = self._sampler
        if self._batch_size == 0:
            raise ValueError("the trajectory has not been evaluated yet: its batch size is unknown")
        if self._traj is not None:
            self._traj.close()
        self._traj = s._new_engine_trajectory(self._batch_size)

    def __call__(self, x):
        """x [N, B, D] -> [N, B, 1]."""
        x = x if type(x).__module__.startswith("torch") else np.asarray(x, dtype=np.float64)
        if len(x.shape) != 3:
            raise ValueError(f"trajectory inputs must be [N, B, D], got shape {tuple(x.shape)}")
        self._ensure(int(x.shape[1]))
        if x.shape[1] == 1:
            return self._traj(x[:, 0, :])[..., None]
        return self._traj(x)[..., None]

    def value_and_gradient(self, x):
        """x [P, B, D] -> (values [P, B], d f_b / d x_b [P, B, D]): what the L-BFGS-B refinement of the
        continuous Thompson-sampling builders consumes (there is no autodiff on this engine)."""
        x = x if type(x).__module__.startswith("torch") else np.asarray(x, dtype=np.float64)
        if len(x.shape) == 2:  # batch-size-one optimizers pass [P, D]: a single trajectory
            val, grad = self.value_and_gradient(x[:, None, :])
            return val[:, 0], grad[:, 0, :]
        if len(x.shape) != 3:
            raise ValueError(f"trajectory inputs must be [P, B, D], got shape {tuple(x.shape)}")
        self._ensure(int(x.shape[1]))
        return self._traj.value_and_gradient(x)

    def argmin_over(self, at):
        """Fused evaluate + arg-min over shared candidates at [N, D] for every trajectory of the
        batch -> (values [B], indices [B]); the [N, B] evaluations never leave the GPU."""
        self._ensure(self._batch_size or 1)
        return self._traj.argmin(at)


class DecoupledTrajectorySampler:
    """Decoupled (RFF prior + canonical update) trajectory sampler for an exact GPR
    (reference sampler.py:594-738).  The RFF basis is drawn once; ``get_trajectory`` returns a
    trajectory with fresh weights, ``resample_trajectory`` redraws weights in place,
    ``update_trajectory`` redraws the basis too (after a model update)."""

    def __init__(self, model, num_features: int = 1000, seed: Optional[int] = None):
        for attr in ("get_kernel", "get_observation_noise", "get_internal_data", "engine"):
            if not hasattr(model, attr):
                raise NotImplementedError(
                    "DecoupledTrajectorySampler only works with models that support get_kernel, "
                    f"get_observation_noise and get_internal_data on this engine; but received {model!r}.")
        if num_features <= 0:
            raise ValueError(f"num_features must be positive, got {num_features}")
        self._model = model
        self._num_features = num_features
        self._rng = make_rng(seed)
        self._resample_basis()

    def __repr__(self) -> str:
        return f"DecoupledTrajectorySampler({self._model!r}, {self._num_features!r})"

    def _resample_basis(self) -> None:
        k = self._model.get_kernel()
        self._W, self._b = sample_rff_basis(k.kind, self._num_features, self._model.engine.d, self._rng)

    def _new_engine_trajectory(self, batch_size: int):
        """Fresh weights: prior weights w [F, B] and noise draws xi [N, B]; the canonical weights v are solved
        on the GPU from the cached factor."""
        w = self._rng.standard_normal((self._num_features, batch_size))
        xi = self._rng.standard_normal((self._model.engine.N, batch_size))
        return self._model.engine.trajectory(self._W, self._b, w, xi)

    def get_trajectory(self) -> decoupled_trajectory:
        return decoupled_trajectory(self)

    def resample_trajectory(self, trajectory: decoupled_trajectory) -> decoupled_trajectory:
        trajectory.resample()
        return trajectory

    def update_trajectory(self, trajectory: decoupled_trajectory) -> decoupled_trajectory:
        self._resample_basis()
        trajectory.resample()
        return trajectory


class RandomFourierFeatureTrajectorySampler(DecoupledTrajectorySampler):
    """Trajectories f(x) = phi(x) . theta + m(x) with theta drawn from the posterior of a Bayesian linear
    model over the F Fourier features (reference sampler.py:452-591): "design space" (an F x F
    factorisation) when F < N, "gram space" (N x N) otherwise -- built, factorised and sampled on the
    GPU (tgp_traj_create_rff).  Same trajectory object / resample / update protocol as the decoupled
    sampler; selected by ``GaussianProcessRegression(..., use_decoupled_sampler=False)``."""

    def __repr__(self) -> str:
        return f"RandomFourierFeatureTrajectorySampler({self._model!r}, {self._num_features!r})"

    def _new_engine_trajectory(self, batch_size: int):
        eps = self._rng.standard_normal((self._num_features, batch_size))
        return self._model.engine.trajectory_rff(self._W, self._b, eps)
