"""Asynchronous rules (reference trieste/acquisition/rule.py: AsynchronousRuleState 403-489, AsynchronousOptimization
492-677, AsynchronousGreedy 680-833).  EXTRAS: outside the hot-path scope of SURVEY.md section 8 (section 2 row 7),
frozen -- host orchestration over the same engine-backed builders, kept because it is tested and works."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Mapping, Optional

import numpy as np

from ..acquisition.function import BatchMonteCarloExpectedImprovement
from ..acquisition.interface import (AcquisitionFunctionBuilder, GreedyAcquisitionFunctionBuilder,
                                     SingleModelAcquisitionBuilder, SingleModelGreedyAcquisitionBuilder)
from ..acquisition.optimizer import automatic_optimizer_selector, batchify_joint
from ..acquisition.rule import AcquisitionRule
from ..data import OBJECTIVE
from ..space import SearchSpace


@dataclass(frozen=True)
class AsynchronousRuleState:
    """Pending points of the asynchronous rules: requested but not yet observed (rule.py:403-489)."""

    pending_points: Optional[np.ndarray] = None

    def __post_init__(self) -> None:
        if self.pending_points is None:
            return
        pts = np.asarray(self.pending_points, dtype=np.float64)
        if pts.ndim != 2:
            raise ValueError(f"Pending points are expected to be a 2D tensor, instead received tensor of shape {pts.shape}")
        object.__setattr__(self, "pending_points", pts)

    @property
    def has_pending_points(self) -> bool:
        return self.pending_points is not None and self.pending_points.size > 0

    def remove_points(self, points_to_remove) -> "AsynchronousRuleState":
        """Drop from the pending points every row present in ``points_to_remove``; a point occurring several times
        among the pending ones loses only its first occurrence per removal (rule.py:426-470)."""
        if not self.has_pending_points:
            return self
        rem = np.asarray(points_to_remove, dtype=np.float64)
        if rem.ndim != 2 or rem.shape[-1] != self.pending_points.shape[-1]:
            raise ValueError(f"Point to remove shall be 1xD where D is the last dimension of pending points. Got "
                             f"{self.pending_points.shape} for pending points and {rem.shape} for other points.")
        pending = self.pending_points
        for point in rem:
            equal = np.all(pending == point, axis=1)
            if equal.any():
                first = int(np.argmax(equal))
                pending = np.concatenate([pending[:first], pending[first + 1:]], axis=0)
        return AsynchronousRuleState(pending)

    def add_pending_points(self, new_points) -> "AsynchronousRuleState":
        """rule.py:472-489."""
        new = np.asarray(new_points, dtype=np.float64)
        if not self.has_pending_points:
            return AsynchronousRuleState(new)
        if new.ndim != 2 or new.shape[-1] != self.pending_points.shape[-1]:
            raise ValueError(f"New points shall be 2D and have same last dimension as pending points. Got "
                             f"{self.pending_points.shape} for pending points and {new.shape} for new points.")
        return AsynchronousRuleState(np.concatenate([self.pending_points, new], axis=0))


def _check_single_objective(models: Mapping, datasets: Optional[Mapping]) -> None:
    if set(models.keys()) != {OBJECTIVE}:
        raise ValueError(f"dict of models must contain the single key {OBJECTIVE}, got keys {models.keys()}")
    if datasets is None or set(datasets.keys()) != {OBJECTIVE}:
        raise ValueError(f"datasets must be provided and contain the single key {OBJECTIVE}")


class AsynchronousOptimization(AcquisitionRule):
    """Asynchronous BO with a *batch* acquisition function (rule.py:492-677): workers return observations one at a
    time, so each call proposes ``num_query_points`` new points given the P points still pending -- the batch
    function is evaluated on [pending; candidate] batches of size P + B and optimised over the candidate part
    only.  ``acquire`` returns a state function ``state -> (new state, points)``; the loops thread the state."""

    def __init__(self, builder=None, optimizer=None, num_query_points: int = 1):
        if num_query_points <= 0:
            raise ValueError(f"Number of query points must be greater than 0, got {num_query_points}")
        if builder is None:
            builder = BatchMonteCarloExpectedImprovement(10_000)
        if optimizer is None:
            optimizer = automatic_optimizer_selector
        if isinstance(builder, SingleModelAcquisitionBuilder):
            builder = builder.using(OBJECTIVE)
        if not isinstance(builder, AcquisitionFunctionBuilder):
            raise TypeError(f"unsupported acquisition builder {builder!r}")
        if num_query_points > 1:  # no need to batchify_joint for a batch of one
            optimizer = batchify_joint(optimizer, num_query_points)
        self._builder = builder
        self._optimizer = optimizer
        self._acquisition_function = None

    def __repr__(self) -> str:
        return f"AsynchronousOptimization({self._builder!r}, {self._optimizer!r})"

    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        _check_single_objective(models, datasets)
        if self._acquisition_function is None:
            self._acquisition_function = self._builder.prepare_acquisition_function(models, datasets=datasets)
        else:
            self._acquisition_function = self._builder.update_acquisition_function(
                self._acquisition_function, models, datasets=datasets)

        def state_func(state: Optional[AsynchronousRuleState]):
            if state is None:
                state = AsynchronousRuleState(None)
            state = state.remove_points(datasets[OBJECTIVE].query_points)
            acquisition_function = self._acquisition_function
            if state.has_pending_points:
                pending = state.pending_points
                base = self._acquisition_function

                def function_with_pending_points(x):  # [N, B, D] -> the batch function on [N, P + B, D]
                    x = np.asarray(x, dtype=np.float64)
                    repeated = np.broadcast_to(pending[None, :, :], (x.shape[0],) + pending.shape)
                    return base(np.concatenate([repeated, x], axis=1))

                acquisition_function = function_with_pending_points
            new_points = self._optimizer(search_space, acquisition_function)
            return state.add_pending_points(new_points), new_points

        return state_func


class AsynchronousGreedy(AcquisitionRule):
    """Asynchronous BO with a *greedy* batch builder (LocalPenalization, Fantasizer, GIBBON; rule.py:680-833): the
    pending points of the state are the builder's pending points; B greedy steps per call, each new point joining
    the pending set."""

    def __init__(self, builder, optimizer=None, num_query_points: int = 1):
        if num_query_points <= 0:
            raise ValueError(f"Number of query points must be greater than 0, got {num_query_points}")
        if builder is None:
            raise ValueError("Please specify an acquisition builder")
        if not isinstance(builder, (GreedyAcquisitionFunctionBuilder, SingleModelGreedyAcquisitionBuilder)):
            raise NotImplementedError(f"Only greedy acquisition strategies are supported, got {type(builder)}")
        if optimizer is None:
            optimizer = automatic_optimizer_selector
        if isinstance(builder, SingleModelGreedyAcquisitionBuilder):
            builder = builder.using(OBJECTIVE)
        self._builder = builder
        self._optimizer = optimizer
        self._acquisition_function = None
        self._num_query_points = num_query_points

    def __repr__(self) -> str:
        return f"AsynchronousGreedy({self._builder!r}, {self._optimizer!r}, {self._num_query_points!r})"

    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        _check_single_objective(models, datasets)

        def state_func(state: Optional[AsynchronousRuleState]):
            if state is None:
                state = AsynchronousRuleState(None)
            state = state.remove_points(datasets[OBJECTIVE].query_points)
            pending = state.pending_points if state.has_pending_points else None
            if self._acquisition_function is None:
                self._acquisition_function = self._builder.prepare_acquisition_function(
                    models, datasets=datasets, pending_points=pending)
            else:
                self._acquisition_function = self._builder.update_acquisition_function(
                    self._acquisition_function, models, datasets=datasets, pending_points=pending)
            new_points_batch = self._optimizer(search_space, self._acquisition_function)
            state = state.add_pending_points(new_points_batch)
            for _ in range(self._num_query_points - 1):  # greedily allocate additional batch elements
                self._acquisition_function = self._builder.update_acquisition_function(
                    self._acquisition_function, models, datasets=datasets, pending_points=state.pending_points,
                    new_optimization_step=False)
                new_point = self._optimizer(search_space, self._acquisition_function)
                state = state.add_pending_points(new_point)
                new_points_batch = np.concatenate([new_points_batch, new_point], axis=0)
            return state, new_points_batch

        return state_func


