"""Search spaces on the hot path (reference trieste/space.py): ``Box`` (sample: 843-867) and
``DiscreteSearchSpace`` (387-505).  Product/tagged/constrained spaces are host-side set algebra
outside the path."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .rng import make_rng


class SearchSpace:
    has_constraints = False

    @property
    def dimension(self) -> int:
        raise NotImplementedError

    def sample(self, num_samples: int, seed: Optional[int] = None) -> np.ndarray:
        raise NotImplementedError


class Box(SearchSpace):
    """Axis-aligned box [lower, upper]."""

    def __init__(self, lower: Sequence[float], upper: Sequence[float]):
        lo = np.asarray(lower, dtype=np.float64).reshape(-1)
        up = np.asarray(upper, dtype=np.float64).reshape(-1)
        if lo.shape != up.shape or lo.size == 0:
            raise ValueError(f"lower and upper must be non-empty and of equal shape, got {lo.shape}, {up.shape}")
        if np.any(lo >= up):
            raise ValueError(f"lower bound must be below upper bound in every dimension, got {lo}, {up}")
        self._lower, self._upper = lo, up

    def __repr__(self) -> str:
        return f"Box({self._lower!r}, {self._upper!r})"

    @property
    def lower(self) -> np.ndarray:
        return self._lower

    @property
    def upper(self) -> np.ndarray:
        return self._upper

    @property
    def dimension(self) -> int:
        return int(self._lower.shape[0])

    def __contains__(self, value) -> bool:
        v = np.asarray(value, dtype=np.float64)
        if v.shape[-1] != self.dimension:
            raise ValueError(f"point must have dimension {self.dimension}, got shape {v.shape}")
        return bool(np.all(v >= self._lower) and np.all(v <= self._upper))

    def sample(self, num_samples: int, seed: Optional[int] = None) -> np.ndarray:
        """``num_samples`` i.i.d. uniform points [num_samples, D].  (TF's own Philox stream cannot be
        reproduced outside TF; the distribution and the seed-reproducibility contract are kept.)"""
        if num_samples < 0:
            raise ValueError(f"num_samples must be non-negative, got {num_samples}")
        rng = make_rng(seed)
        return rng.uniform(self._lower, self._upper, size=(num_samples, self.dimension))

    def sample_halton(self, num_samples: int, seed: Optional[int] = None) -> np.ndarray:
        """``num_samples`` points of a randomised Halton sequence [num_samples, D] (space.py:869-897: tfp's
        ``sample_halton_sequence``, randomised unless seeded; here scipy's scrambled Halton generator -- TFP's
        permutation stream cannot be reproduced outside TF, the low-discrepancy and seed-reproducibility
        contracts are kept)."""
        if num_samples < 0:
            raise ValueError(f"num_samples must be non-negative, got {num_samples}")
        if num_samples == 0:
            return np.zeros((0, self.dimension))
        from scipy.stats import qmc

        gen = qmc.Halton(d=self.dimension, scramble=True, seed=make_rng(seed))
        return (self._upper - self._lower) * gen.random(num_samples) + self._lower

    def sample_sobol(self, num_samples: int, skip: Optional[int] = None) -> np.ndarray:
        """``num_samples`` points of the (unscrambled) Sobol sequence after skipping ``skip`` points
        [num_samples, D] (space.py:899-917: ``tf.math.sobol_sample``; same direction numbers, TF's stream starts
        after the all-zero point).  ``skip=None`` draws a random skip below 2^16 like the reference."""
        if num_samples < 0:
            raise ValueError(f"num_samples must be non-negative, got {num_samples}")
        if num_samples == 0:
            return np.zeros((0, self.dimension))
        import warnings

        from scipy.stats import qmc

        if skip is None:
            skip = int(make_rng().integers(0, 2 ** 16))
        gen = qmc.Sobol(d=self.dimension, scramble=False)
        gen.fast_forward(int(skip) + 1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # scipy warns when num_samples is not a power of two
            pts = gen.random(num_samples)
        return (self._upper - self._lower) * pts + self._lower

    def sample_device(self, engine, num_samples: int, seed: int = 0, first: int = 0):
        """The same distribution generated on the GPU (Philox4x32-10): a torch CUDA tensor
        [num_samples, D]; row ``first + i`` depends only on (seed, first + i), so shards agree."""
        return engine.sample_box(seed, first, num_samples, self._lower, self._upper)

    def __pow__(self, other: int) -> "Box":
        if other < 1:
            raise ValueError(f"exponent must be strictly positive, got {other}")
        return Box(np.tile(self._lower, other), np.tile(self._upper, other))

    def __mul__(self, other: "Box") -> "Box":
        return Box(np.concatenate([self._lower, other._lower]), np.concatenate([self._upper, other._upper]))


class DiscreteSearchSpace(SearchSpace):
    """A finite table of points [N, D]."""

    def __init__(self, points):
        p = np.asarray(points, dtype=np.float64)
        if p.ndim != 2 or p.shape[0] == 0:
            raise ValueError(f"points must be a non-empty [N, D] array, got shape {p.shape}")
        self._points = p

    def __repr__(self) -> str:
        return f"DiscreteSearchSpace({self._points!r})"

    @property
    def points(self) -> np.ndarray:
        return self._points

    @property
    def lower(self) -> np.ndarray:
        return self._points.min(axis=0)

    @property
    def upper(self) -> np.ndarray:
        return self._points.max(axis=0)

    @property
    def dimension(self) -> int:
        return int(self._points.shape[1])

    def __contains__(self, value) -> bool:
        v = np.asarray(value, dtype=np.float64)
        return bool(np.any(np.all(self._points == v, axis=1)))

    def sample(self, num_samples: int, seed: Optional[int] = None) -> np.ndarray:
        """Uniform sampling of the table; if fewer points than requested, all of them
        (reference space.py:470-489)."""
        if num_samples < 0:
            raise ValueError(f"num_samples must be non-negative, got {num_samples}")
        if num_samples == 0:
            return self._points[:0]
        rng = make_rng(seed)
        n = self._points.shape[0]
        idx = rng.permutation(n)[: min(num_samples, n)]
        return self._points[idx]
