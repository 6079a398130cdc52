"""Acquisition rules of the hot path (reference trieste/acquisition/rule.py): AcquisitionRule
(109-190), EfficientGlobalOptimization (209-399), RandomSampling (836-876), DiscreteThompsonSampling (879-994).
The asynchronous rules live in trieste_amd/extras (outside the SURVEY section 8 scope)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Mapping, Optional

import numpy as np

from ..data import OBJECTIVE
from ..space import SearchSpace
from .function import BatchMonteCarloExpectedImprovement, ExpectedImprovement
from .interface import (AcquisitionFunctionBuilder, GreedyAcquisitionFunctionBuilder, SingleModelAcquisitionBuilder,
                        SingleModelGreedyAcquisitionBuilder, VectorizedAcquisitionFunctionBuilder)
from .optimizer import _fresh_seed, automatic_optimizer_selector, batchify_joint, batchify_vectorize
from .sampler import ExactThompsonSampler, ThompsonSampler
from .utils import select_nth_output


class AcquisitionRule(ABC):
    """``acquire(search_space, models, datasets) -> query points`` (rule.py:126-147)."""

    @abstractmethod
    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        ...

    def acquire_single(self, search_space: SearchSpace, model, dataset=None):
        """Convenience for one model / dataset under the OBJECTIVE tag (rule.py:149-170)."""
        return self.acquire(search_space, {OBJECTIVE: model},
                            datasets=None if dataset is None else {OBJECTIVE: dataset})


class EfficientGlobalOptimization(AcquisitionRule):
    """Efficient Global Optimization: build/update the acquisition function, maximise it with the
    optimizer (rule.py:209-399).  ``num_query_points > 1``: a joint batch builder wraps the
    optimizer with :func:`batchify_joint`, a vectorized builder with :func:`batchify_vectorize`, a
    greedy builder is re-updated with the pending points and re-optimised per batch element."""

    def __init__(self, builder=None, optimizer=None, num_query_points: int = 1,
                 initial_acquisition_function=None):
        if num_query_points <= 0:
            raise ValueError(f"Number of query points must be greater than 0, got {num_query_points}")
        if builder is None:
            if num_query_points == 1:
                builder = ExpectedImprovement()
            else:
                raise ValueError("Need to specify a batch acquisition function when number of query points is "
                                 "greater than 1")
        if optimizer is None:
            optimizer = automatic_optimizer_selector
        if isinstance(builder, (SingleModelAcquisitionBuilder, SingleModelGreedyAcquisitionBuilder)):
            builder = builder.using(OBJECTIVE)
        if not isinstance(builder, (AcquisitionFunctionBuilder, GreedyAcquisitionFunctionBuilder)):
            raise TypeError(f"unsupported acquisition builder {builder!r}")
        self._base_optimizer = optimizer  # before any batch wrapping (trust-region rules re-use it per region)
        if num_query_points > 1:
            if isinstance(builder, VectorizedAcquisitionFunctionBuilder):
                optimizer = batchify_vectorize(optimizer, num_query_points)  # batch elements independently
            elif isinstance(builder, AcquisitionFunctionBuilder):
                optimizer = batchify_joint(optimizer, num_query_points)  # batch elements jointly
            # greedy builders: sequentially, in acquire()
        self._builder = builder
        self._optimizer = optimizer
        self._num_query_points = num_query_points
        self._acquisition_function = initial_acquisition_function

    def __repr__(self) -> str:
        return f"EfficientGlobalOptimization({self._builder!r}, {self._optimizer!r}, {self._num_query_points!r})"

    @property
    def acquisition_function(self):
        """The current acquisition function, updated last time :meth:`acquire` was called."""
        return self._acquisition_function

    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        if self._acquisition_function is None:
            self._acquisition_function = self._builder.prepare_acquisition_function(models, datasets=datasets)
        else:
            self._acquisition_function = self._builder.update_acquisition_function(
                self._acquisition_function, models, datasets=datasets)
        points = self._optimizer(search_space, self._acquisition_function)
        if isinstance(self._builder, GreedyAcquisitionFunctionBuilder):
            for _ in range(self._num_query_points - 1):  # greedily allocate the remaining batch elements
                self._acquisition_function = self._builder.update_acquisition_function(
                    self._acquisition_function, models, datasets=datasets, pending_points=points,
                    new_optimization_step=False)
                chosen_point = self._optimizer(search_space, self._acquisition_function)
                points = np.concatenate([points, chosen_point], axis=0)
        return points


class RandomSampling(AcquisitionRule):
    """Uniformly random query points (rule.py:836-876)."""

    def __init__(self, num_query_points: int = 1):
        if num_query_points <= 0:
            raise ValueError(f"Number of query points must be greater than 0, got {num_query_points}")
        self._num_query_points = num_query_points

    def __repr__(self) -> str:
        return f"RandomSampling({self._num_query_points!r})"

    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        return search_space.sample(self._num_query_points)


class DiscreteThompsonSampling(AcquisitionRule):
    """Thompson sampling over a random discretisation of the space (rule.py:879-994): sample
    ``num_search_space_samples`` candidates, return the minimisers of ``num_query_points`` posterior
    draws."""

    def __init__(self, num_search_space_samples: int, num_query_points: int,
                 thompson_sampler: Optional[ThompsonSampler] = None, select_output=select_nth_output,
                 seed: Optional[int] = None, on_device: bool = True):
        if not num_search_space_samples > 0:
            raise ValueError(f"Search space must be greater than 0, got {num_search_space_samples}")
        if not num_query_points > 0:
            raise ValueError(f"Number of query points must be greater than 0, got {num_query_points}")
        if thompson_sampler is not None:
            if thompson_sampler.sample_min_value:
                raise ValueError("Thompson sampling requires a thompson_sampler that samples minimizers, not just "
                                 "minimum values. However the passed sampler has sample_min_value=True.")
        else:
            thompson_sampler = ExactThompsonSampler(sample_min_value=False)
        self._thompson_sampler = thompson_sampler
        self._num_search_space_samples = num_search_space_samples
        self._num_query_points = num_query_points
        self._select_output = select_output
        self._seed = seed
        self._on_device = on_device

    def __repr__(self) -> str:
        return (f"DiscreteThompsonSampling({self._num_search_space_samples!r}, {self._num_query_points!r}, "
                f"{self._thompson_sampler!r}, {self._select_output!r})")

    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        if set(models.keys()) != {OBJECTIVE}:
            raise ValueError(f"dict of models must contain the single key {OBJECTIVE}, got keys {models.keys()}")
        if datasets is None or set(datasets.keys()) != {OBJECTIVE}:
            raise ValueError(f"datasets must be provided and contain the single key {OBJECTIVE}")
        model = models[OBJECTIVE]
        eng = getattr(model, "engine", None)
        if self._on_device and eng is not None and hasattr(search_space, "sample_device") and hasattr(eng, "sample_box"):
            query_points = search_space.sample_device(eng, self._num_search_space_samples,
                                                      seed=_fresh_seed() if self._seed is None else self._seed)
        else:
            query_points = search_space.sample(self._num_search_space_samples, seed=self._seed)
        return self._thompson_sampler.sample(model, self._num_query_points, query_points,
                                             select_output=self._select_output)
