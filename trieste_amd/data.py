"""Dataset container (reference trieste/data.py:25-110) and the OBJECTIVE tag (observer.py)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

OBJECTIVE = "OBJECTIVE"
"""Tag of the objective model/dataset (reference trieste/observer.py)."""


@dataclass(frozen=True)
class Dataset:
    """query_points [..., N, D] and observations [..., N, L] with matching leading shapes."""

    query_points: np.ndarray
    observations: np.ndarray

    def __post_init__(self) -> None:
        q = np.asarray(self.query_points, dtype=np.float64)
        o = np.asarray(self.observations, dtype=np.float64)
        object.__setattr__(self, "query_points", q)
        object.__setattr__(self, "observations", o)
        if q.ndim < 2 or o.ndim < 2:
            raise ValueError(f"query_points and observations must have rank >= 2, got {q.shape}, {o.shape}")
        if 0 in (q.shape[-1], o.shape[-1]):
            raise ValueError(f"query_points and observations cannot have dimension 0, got shapes {q.shape} and {o.shape}.")
        if q.shape[:-1] != o.shape[:-1]:
            raise ValueError(f"Leading shapes of query_points and observations must match. Got shapes {q.shape}, {o.shape}.")

    def __add__(self, rhs: "Dataset") -> "Dataset":
        if self.query_points.shape[1:] != rhs.query_points.shape[1:] or \
                self.observations.shape[1:] != rhs.observations.shape[1:]:
            raise ValueError("datasets can only differ in their zeroth dimension")
        return Dataset(np.concatenate([self.query_points, rhs.query_points], axis=0),
                       np.concatenate([self.observations, rhs.observations], axis=0))

    def __len__(self) -> int:
        return int(self.observations.shape[0])

    def __deepcopy__(self, memo):
        return self

    def astuple(self):
        return self.query_points, self.observations
