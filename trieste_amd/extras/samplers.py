"""EXTRAS (out of scope per SURVEY.md section 2 row 22; moved here from ``trieste_amd.acquisition.sampler`` in round 6): the
Gumbel min-value sampler of the entropy-search family (reference acquisition/sampler.py:126-212)."""
from __future__ import annotations

import numpy as np

from ..acquisition.sampler import ThompsonSampler, _check, _is_torch
from ..acquisition.utils import select_nth_output


class GumbelSampler(ThompsonSampler):
    r"""Approximate samples of the objective's minimum value :math:`y^*` (Wang & Jegelka 2017; sampler.py:126-212):
    the empirical cdf :math:`\Pr(y^* < y) = 1 - \prod_i \Phi(-(y - \mu_i) / \sigma_i)` over the grid is matched at
    its quartiles by a Gumbel distribution :math:`1 - e^{-e^{(y - a) / b}}`, which is then sampled by inversion.
    The model's observation-space predictions at the grid come from the engine (one sweep); the two scalar
    bisections and the S inversions are host arithmetic on that [N] vector, as scipy calls in the reference."""

    def __init__(self, sample_min_value: bool = False):
        if not sample_min_value:
            raise ValueError(f"Gumbel samplers can only sample a function's minimal value, however received "
                             f"sample_min_value={sample_min_value}")
        super().__init__(sample_min_value)

    def sample(self, model, sample_size: int, at, select_output=select_nth_output):
        _check(sample_size, at)
        from scipy.optimize import bisect
        from scipy.special import log_ndtr

        from ..rng import make_rng

        host = at.cpu().numpy() if _is_torch(at) else np.asarray(at)
        fmean, fvar = model.predict_y(host) if hasattr(model, "predict_y") else model.predict(host)
        fmean = np.asarray(fmean, dtype=np.float64).reshape(-1)
        fsd = np.sqrt(np.asarray(fvar, dtype=np.float64).reshape(-1))

        def probf(y: float) -> float:  # empirical cdf Pr(y* < y)
            return 1.0 - float(np.exp(np.sum(log_ndtr(-(y - fmean) / fsd))))

        left, right = float(np.min(fmean - 5.0 * fsd)), float(np.max(fmean + 5.0 * fsd))
        q1, q2 = (bisect(lambda y: probf(y) - val, left, right, maxiter=10000) for val in (0.25, 0.75))
        l1, l2 = np.log(np.log(4.0 / 3.0)), np.log(np.log(4.0))
        b = (q1 - q2) / (l1 - l2)
        a = (q2 * l1 - q1 * l2) / (l1 - l2)
        uniform_samples = make_rng().uniform(size=int(sample_size))
        return (np.log(-np.log(1.0 - uniform_samples)) * b + a)[:, None]  # [S, 1]
