"""Thompson samplers over a discrete candidate set (reference trieste/acquisition/sampler.py:
ThompsonSampler 40-76, ExactThompsonSampler 79-123, ThompsonSamplerFromTrajectory 215-273)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .utils import select_nth_output


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class ThompsonSampler(ABC):
    def __init__(self, sample_min_value: bool = False):
        self._sample_min_value = sample_min_value

    @property
    def sample_min_value(self) -> bool:
        return self._sample_min_value

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self._sample_min_value})"

    @abstractmethod
    def sample(self, model, sample_size: int, at, select_output=select_nth_output):
        ...


def _check(sample_size, at):
    if sample_size <= 0:
        raise ValueError(f"sample_size must be positive, got {sample_size}")
    if len(at.shape) != 2:
        raise ValueError(f"at must be [N, D], got shape {tuple(at.shape)}")


def _gather(at, idx):
    idx = np.asarray(idx)
    if _is_torch(at):
        import torch

        return at[torch.from_numpy(idx).to(at.device)].cpu().numpy()
    return np.asarray(at)[idx]


class ExactThompsonSampler(ThompsonSampler):
    """Exact Thompson samples: joint posterior samples at ALL candidates, then per-sample arg-min
    (sampler.py:79-123).  O(N^3) in the number of candidates -- the covariance assembly and its
    factorisation run on the GPU (``model.sample`` -> tgp_sample_joint)."""

    def sample(self, model, sample_size: int, at, select_output=select_nth_output):
        _check(sample_size, at)
        host = at.cpu().numpy() if _is_torch(at) else np.asarray(at)
        samples = select_output(np.asarray(model.sample(host, sample_size)))  # [S, N]
        if self._sample_min_value:
            return np.min(samples, axis=1, keepdims=True)
        return _gather(at, np.argmin(samples, axis=1))


class ThompsonSamplerFromTrajectory(ThompsonSampler):
    """Approximate Thompson samples: minimise ``sample_size`` trajectories of the model's trajectory
    sampler over the candidates (sampler.py:215-273).  The trajectories share one RFF basis (as in
    the reference, which draws the basis once per sampler) and are evaluated together, fused with
    the per-trajectory arg-min, on the GPU."""

    MAX_BATCH = 16

    def sample(self, model, sample_size: int, at, select_output=select_nth_output):
        _check(sample_size, at)
        if not hasattr(model, "trajectory_sampler"):
            raise ValueError("Thompson sampling from trajectory only supports models with a trajectory_sampler "
                             f"method; received {model!r}")
        sampler = model.trajectory_sampler()
        vals, idxs = [], []
        done = 0
        while done < sample_size:
            b = min(self.MAX_BATCH, sample_size - done)
            traj = sampler.get_trajectory()
            traj._batch_size = b
            traj.resample()
            v, i = traj.argmin_over(at)
            vals.append(np.asarray(v))
            idxs.append(np.asarray(i))
            done += b
        vals, idxs = np.concatenate(vals), np.concatenate(idxs)
        if self._sample_min_value:
            return vals[:, None]
        return _gather(at, idxs)
