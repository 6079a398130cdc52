"""Bayesian optimisation loop (reference trieste/bayesian_optimizer.py:570-883): per step
acquire -> observe -> datasets += new -> model.update -> model.optimize, with the whole step
wrapped so a failure (e.g. a non-positive-definite Cholesky) is returned as ``Err`` together with
the history so far (855-875) instead of propagating."""
from __future__ import annotations

import copy
import traceback
import numpy as np
from dataclasses import dataclass, field
from typing import Callable, List, Mapping, Optional, Union

from .acquisition.rule import AcquisitionRule, EfficientGlobalOptimization
from .data import OBJECTIVE, Dataset
from .space import SearchSpace


@dataclass
class Ok:
    value: object

    is_ok = True
    is_err = False

    def unwrap(self):
        return self.value


@dataclass
class Err:
    error: Exception

    is_ok = False
    is_err = True

    def unwrap(self):
        raise self.error


@dataclass
class Record:
    datasets: Mapping
    models: Mapping
    acquisition_state: object = None

    @property
    def dataset(self) -> Dataset:
        return next(iter(self.datasets.values()))

    @property
    def model(self):
        return next(iter(self.models.values()))


@dataclass
class OptimizationResult:
    final_result: Union[Ok, Err]
    history: List[Record] = field(default_factory=list)

    def try_get_final_datasets(self) -> Mapping:
        return self.final_result.unwrap().datasets

    def try_get_final_dataset(self) -> Dataset:
        return self.final_result.unwrap().dataset

    def try_get_final_models(self) -> Mapping:
        return self.final_result.unwrap().models

    def try_get_final_model(self):
        return self.final_result.unwrap().model

    def try_get_optimal_point(self):
        """(best query point [D], its observation [1], its index) of a single-objective run
        (bayesian_optimizer.py:266-284)."""
        dataset = self.try_get_final_dataset()
        obs = np.asarray(dataset.observations)
        if obs.ndim != 2 or obs.shape[1] != 1:
            raise ValueError("Expected a single objective")
        arg_min_idx = int(np.argmin(obs[:, 0]))
        return dataset.query_points[arg_min_idx], obs[arg_min_idx], arg_min_idx


def stop_at_minimum(minimum=None, minimizers=None, minimum_atol: float = 0, minimum_rtol: float = 0.05,
                    minimizers_atol: float = 0, minimizers_rtol: float = 0.05, objective_tag=OBJECTIVE,
                    minimum_step_number: Optional[int] = None) -> Callable:
    """An early-stop callback that ends a BO loop once the best observation is close to ``minimum`` [1] and / or the
    best query point is close to one of the ``minimizers`` [N, D] (bayesian_optimizer.py:1160-1207).  The step
    number the reference reads from its logging module is counted by the callback itself (one call per step)."""
    calls = {"n": 0}

    def early_stop_callback(datasets: Mapping, _models: Mapping, _acquisition_state) -> bool:
        calls["n"] += 1
        if minimum_step_number is not None and calls["n"] < minimum_step_number:
            return False
        dataset = datasets[objective_tag]
        obs = np.asarray(dataset.observations)
        arg_min_idx = int(np.argmin(obs[:, 0]))
        if minimum is not None and not np.all(np.isclose(obs[arg_min_idx], minimum, atol=minimum_atol, rtol=minimum_rtol)):
            return False
        if minimizers is not None:
            close_x = np.isclose(np.asarray(dataset.query_points)[arg_min_idx], np.asarray(minimizers),
                                 atol=minimizers_atol, rtol=minimizers_rtol)
            if not np.any(np.all(close_x, axis=-1)):
                return False
        return True

    return early_stop_callback


def _as_map(x):
    return x if isinstance(x, Mapping) else {OBJECTIVE: x}


class BayesianOptimizer:
    def __init__(self, observer: Callable, search_space: SearchSpace):
        self._observer = observer
        self._search_space = search_space

    def __repr__(self) -> str:
        return f"BayesianOptimizer({self._observer!r}, {self._search_space!r})"

    def optimize(self, num_steps: int, datasets, models, acquisition_rule: Optional[AcquisitionRule] = None,
                 acquisition_state=None, *,
                 track_state: bool = True, fit_model: bool = True, fit_initial_model: bool = True,
                 early_stop_callback: Optional[Callable] = None) -> OptimizationResult:
        datasets = dict(_as_map(datasets))
        models = dict(_as_map(models))
        if num_steps < 0:
            raise ValueError(f"num_steps must be at least 0, got {num_steps}")
        if datasets.keys() != models.keys():
            raise ValueError(f"datasets and models should contain the same keys. Got {datasets.keys()} and "
                             f"{models.keys()} respectively.")
        if not datasets:
            raise ValueError("dicts of datasets and models must be populated.")
        if acquisition_rule is None:
            if datasets.keys() != {OBJECTIVE}:
                raise ValueError(f"Default acquisition rule EfficientGlobalOptimization requires tag {OBJECTIVE!r}, "
                                 f"got keys {datasets.keys()}")
            acquisition_rule = EfficientGlobalOptimization()
        history: List[Record] = []

        def filter_datasets(state):  # stateful rules (trust regions) update their regions from the newest data
            hook = getattr(acquisition_rule, "filter_datasets", None)
            if hook is None:
                return state
            filtered = hook(models, datasets)
            if callable(filtered):
                state, _ = filtered(state)
            return state

        for step in range(1, num_steps + 1):
            try:
                if track_state:  # per-step copies, like the reference (bayesian_optimizer.py:745-760)
                    history.append(Record(dict(datasets), copy.deepcopy(models), copy.deepcopy(acquisition_state)))
                if step == 1:
                    if hasattr(acquisition_rule, "initialize_subspaces"):
                        acquisition_rule.initialize_subspaces(self._search_space)
                    acquisition_state = filter_datasets(acquisition_state)
                if step == 1 and fit_model and fit_initial_model:
                    for tag, model in models.items():
                        model.update(datasets[tag])
                        model.optimize(datasets[tag])
                points = acquisition_rule.acquire(self._search_space, models, datasets=datasets)
                if callable(points):  # stateful rule (bayesian_optimizer.py:796-800)
                    acquisition_state, points = points(acquisition_state)
                points = np.asarray(points)
                if points.ndim == 3:  # one point per region [N, V, D]: the observer sees the flat batch
                    points = points.reshape(-1, points.shape[-1])
                observed = _as_map(self._observer(points))
                datasets = {tag: datasets[tag] + observed[tag] for tag in datasets}
                acquisition_state = filter_datasets(acquisition_state)
                if fit_model:  # fit_model=False: the caller manages the models (bayesian_optimizer.py:828-834)
                    for tag, model in models.items():
                        model.update(datasets[tag])
                        model.optimize(datasets[tag])
                if early_stop_callback is not None and early_stop_callback(datasets, models, acquisition_state):
                    break
            except Exception as error:  # noqa: BLE001 -- the reference returns Err for any failure
                traceback.print_exc()
                return OptimizationResult(Err(error), history)
        return OptimizationResult(Ok(Record(datasets, models, acquisition_state)), history)
