"""Continuous Thompson sampling builders (reference
trieste/acquisition/function/continuous_thompson_sampling.py:30-245): acquisition functions that are
the NEGATIVES of decoupled posterior trajectories, maximised by the (gradient-based) optimizers to
find the trajectories' minimisers.  Values and gradients come from libtgp's trajectory kernels
(tgp_traj_eval / tgp_traj_value_grad)."""
from __future__ import annotations

from typing import Callable, Optional

from ..data import Dataset
from .interface import SingleModelGreedyAcquisitionBuilder, SingleModelVectorizedAcquisitionBuilder
from .utils import select_nth_output


def negate_trajectory_function(function, select_output: Optional[Callable] = None):
    """Negate a trajectory (and select its output) so that maximisers find its minimisers
    (continuous_thompson_sampling.py:188-245).  Like the reference, the trajectory OBJECT is kept --
    its class is swapped for a subclass with a negated ``__call__`` -- so that ``resample`` /
    ``update`` still act on it in place."""
    base = type(function)
    if getattr(base, "_is_negated_trajectory", False):
        return function

    class NegatedTrajectory(base):
        _is_negated_trajectory = True

        def __call__(self, x):  # [N, B, D] -> [N, B]
            out = base.__call__(self, x)
            return -1.0 * (select_output(out) if select_output is not None else out)

        def value_and_gradient(self, x):
            val, grad = base.value_and_gradient(self, x)
            return -1.0 * val, -1.0 * grad

    function.__class__ = NegatedTrajectory
    return function


def _require_sampler(model):
    if not hasattr(model, "trajectory_sampler"):
        raise ValueError("Thompson sampling from trajectory only supports models with a trajectory_sampler method; "
                         f"received {model!r}")
    return model.trajectory_sampler()


class GreedyContinuousThompsonSampling(SingleModelGreedyAcquisitionBuilder):
    """One negated trajectory per batch element, drawn sequentially
    (continuous_thompson_sampling.py:30-106)."""

    def __init__(self, select_output: Callable = select_nth_output):
        self._select_output = select_output

    def __repr__(self) -> str:
        return f"GreedyContinuousThompsonSampling({self._select_output!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None, pending_points=None):
        self._trajectory_sampler = _require_sampler(model)
        function = self._trajectory_sampler.get_trajectory()
        return negate_trajectory_function(function, self._select_output)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None, pending_points=None,
                                    new_optimization_step: bool = True):
        if new_optimization_step:  # update the sampler (new basis) and resample the trajectory
            new_function = self._trajectory_sampler.update_trajectory(function)
        else:  # same step: only fresh weights
            new_function = self._trajectory_sampler.resample_trajectory(function)
        if new_function is not function:
            function = negate_trajectory_function(new_function, self._select_output)
        return function


class ParallelContinuousThompsonSampling(SingleModelVectorizedAcquisitionBuilder):
    """A batch of negated trajectories optimised in parallel, one per batch element
    (continuous_thompson_sampling.py:109-180)."""

    def __init__(self, select_output: Callable = select_nth_output):
        self._select_output = select_output

    def __repr__(self) -> str:
        return f"ParallelContinuousThompsonSampling({self._select_output!r})"

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        self._trajectory_sampler = _require_sampler(model)
        self._trajectory = self._trajectory_sampler.get_trajectory()
        self._negated_trajectory = negate_trajectory_function(self._trajectory, self._select_output)
        return self._negated_trajectory

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if function is not self._negated_trajectory:
            raise ValueError("Wrong trajectory function passed into update_acquisition_function")
        new_function = self._trajectory_sampler.update_trajectory(self._trajectory)
        if new_function is not self._trajectory:  # negate again if it was not modified in place
            self._trajectory = new_function
            self._negated_trajectory = negate_trajectory_function(new_function, self._select_output)
        return self._negated_trajectory
