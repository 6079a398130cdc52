"""Continuous Thompson sampling on the engine's trajectories.

The reference's builders (trieste/acquisition/function/continuous_thompson_sampling.py:30-245) hand the optimizer
the NEGATIVE of a posterior trajectory so that a maximiser finds the trajectory's minimiser.  Here a trajectory
is a device object (``tgp_traj``: RFF features + canonical kernel sums, values by ``tgp_traj_eval``, analytic
gradients by ``tgp_traj_value_grad``), so the negation is a thin VIEW over it -- :class:`NegatedTrajectory` --
instead of the reference's in-place class swap: the view owns nothing, flips the sign of values and gradients on
the way out, and keeps pointing at the same trajectory while the sampler redraws its weights or basis in place.
"""
from __future__ import annotations

from typing import Callable, Optional

from ..data import Dataset
from .interface import SingleModelGreedyAcquisitionBuilder, SingleModelVectorizedAcquisitionBuilder
from .utils import select_nth_output


class NegatedTrajectory:
    """``x [N, B, D] -> -select_output(trajectory(x))``; ``value_and_gradient`` flips both signs exactly once,
    for [P, B, D] and for the [P, D] points batch-size-one optimizers pass."""

    def __init__(self, trajectory, select_output: Optional[Callable] = None):
        self.trajectory = trajectory
        self._select_output = select_output

    def __call__(self, x):
        out = self.trajectory(x)
        return -(out if self._select_output is None else self._select_output(out))

    def value_and_gradient(self, x):
        val, grad = self.trajectory.value_and_gradient(x)
        return -val, -grad


def negate_trajectory_function(function, select_output: Optional[Callable] = None) -> NegatedTrajectory:
    """The maximisable view of a trajectory (continuous_thompson_sampling.py:188-245); a view is returned as is."""
    return function if isinstance(function, NegatedTrajectory) else NegatedTrajectory(function, select_output)


class _TrajectoryViewBuilder:
    """What the two builders share: a trajectory sampler taken from the model, one view handed out, and a refresh
    of the trajectory BEHIND that view (fresh basis + weights after a model update; fresh weights only inside a
    step) so the optimizer-facing object never changes identity."""

    def __init__(self, select_output: Callable = select_nth_output):
        self._select_output = select_output
        self._sampler = None
        self._view: Optional[NegatedTrajectory] = None

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self._select_output!r})"

    def _fresh_view(self, model) -> NegatedTrajectory:
        if not hasattr(model, "trajectory_sampler"):
            raise ValueError("Thompson sampling from trajectory only supports models with a trajectory_sampler "
                             f"method; received {model!r}")
        self._sampler = model.trajectory_sampler()
        self._view = NegatedTrajectory(self._sampler.get_trajectory(), self._select_output)
        return self._view

    def _refresh(self, view: NegatedTrajectory, new_basis: bool) -> NegatedTrajectory:
        redraw = self._sampler.update_trajectory if new_basis else self._sampler.resample_trajectory
        view.trajectory = redraw(view.trajectory)  # in place on this engine; rebinding covers samplers that copy
        return view


class GreedyContinuousThompsonSampling(_TrajectoryViewBuilder, SingleModelGreedyAcquisitionBuilder):
    """One trajectory at a time (continuous_thompson_sampling.py:30-106): EGO's greedy loop asks for an update per
    batch element -- new weights inside a step, a new basis as well when a new optimisation step begins."""

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None, pending_points=None):
        return self._fresh_view(model)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None, pending_points=None,
                                    new_optimization_step: bool = True):
        view = negate_trajectory_function(function, self._select_output)
        return self._refresh(view, new_basis=new_optimization_step)


class ParallelContinuousThompsonSampling(_TrajectoryViewBuilder, SingleModelVectorizedAcquisitionBuilder):
    """B trajectories optimised together, column b at its own point (continuous_thompson_sampling.py:109-180): the
    engine evaluates all B columns and their gradients in one launch."""

    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        return self._fresh_view(model)

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        if function is not self._view:
            raise ValueError("Wrong trajectory function passed into update_acquisition_function")
        return self._refresh(self._view, new_basis=True)
