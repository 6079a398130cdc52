"""Ask-Tell loop (reference trieste/ask_tell_optimization.py:595-729), the caller side of the hot
path: ``ask`` -> rule.acquire(...), ``tell`` -> datasets += new data; model.update; model.optimize.
Only what the loop needs is restated; checkpoint records and TensorBoard logging are out of scope
(SURVEY.md section 2 rows 15-16)."""
from __future__ import annotations

from typing import Mapping, Optional, Union

from .acquisition.rule import AcquisitionRule, EfficientGlobalOptimization
from .data import OBJECTIVE, Dataset
from .space import SearchSpace


def _as_map(x):
    return x if isinstance(x, Mapping) else {OBJECTIVE: x}


class AskTellOptimizer:
    def __init__(self, search_space: SearchSpace, datasets: Union[Mapping, Dataset], models,
                 acquisition_rule: Optional[AcquisitionRule] = None, acquisition_state=None, *,
                 fit_model: bool = True, track_data: bool = True):
        self._search_space = search_space
        self._datasets = dict(_as_map(datasets))
        self._models = dict(_as_map(models))
        if not self._datasets or not self._models:
            raise ValueError("dicts of datasets and models must be populated.")
        if self._datasets.keys() != self._models.keys():
            raise ValueError(f"datasets and models should contain the same keys. Got {self._datasets.keys()} and "
                             f"{self._models.keys()} respectively.")
        if acquisition_rule is None:
            if self._datasets.keys() != {OBJECTIVE}:
                raise ValueError(f"Default acquisition rule EfficientGlobalOptimization requires tag {OBJECTIVE!r}, "
                                 f"got keys {self._datasets.keys()}")
            acquisition_rule = EfficientGlobalOptimization()
        self._acquisition_rule = acquisition_rule
        self._acquisition_state = acquisition_state  # of stateful rules (ask_tell_optimization.py:237)
        self._track_data = track_data
        # rules with regions (trust regions) set them up against the search space and see every new data set
        # (ask_tell_optimization.py:316-330; bayesian_optimizer.py:755-770)
        if hasattr(acquisition_rule, "initialize_subspaces"):
            acquisition_rule.initialize_subspaces(search_space)
        self._filter_datasets()
        if fit_model:  # the INITIAL fit only (ask_tell_optimization.py:221, 333-340); `tell` always updates
            for tag, model in self._models.items():
                self.update_model(model, self._datasets[tag])

    def update_model(self, model, dataset: Dataset) -> None:
        """Refresh one model with its data set: update, then train (ask_tell_optimization.py:744-746)."""
        model.update(dataset)
        model.optimize(dataset)

    def __repr__(self) -> str:
        return (f"AskTellOptimizer({self._search_space!r}, {self._datasets!r}, {self._models!r}, "
                f"{self._acquisition_rule!r})")

    @property
    def datasets(self) -> Mapping:
        return self._datasets

    @property
    def dataset(self) -> Dataset:
        if len(self._datasets) == 1:
            return next(iter(self._datasets.values()))
        raise ValueError(f"Expected a single dataset, found {len(self._datasets)}")

    @property
    def models(self) -> Mapping:
        return self._models

    @property
    def model(self):
        if len(self._models) == 1:
            return next(iter(self._models.values()))
        raise ValueError(f"Expected a single model, found {len(self._models)}")

    def _filter_datasets(self) -> None:
        """Let the rule look at (models, datasets): a stateful rule returns ``state -> (state, datasets)`` and
        updates its regions there.  Global datasets come back unchanged."""
        hook = getattr(self._acquisition_rule, "filter_datasets", None)
        if hook is None:
            return
        filtered = hook(self._models, self._datasets)
        if callable(filtered):
            self._acquisition_state, filtered = filtered(self._acquisition_state)

    @property
    def acquisition_state(self):
        """The state of a stateful acquisition rule (ask_tell_optimization.py:431-433)."""
        return self._acquisition_state

    def ask(self):
        """Suggest the next query point(s) (ask_tell_optimization.py:595-640).  A stateful rule returns a
        function ``state -> (new state, points)``; the optimizer threads its state through it (:613-618)."""
        points_or_stateful = self._acquisition_rule.acquire(self._search_space, self._models, datasets=self._datasets)
        if callable(points_or_stateful):
            self._acquisition_state, query_points = points_or_stateful(self._acquisition_state)
            return query_points
        return points_or_stateful

    def tell(self, new_data: Union[Mapping, Dataset]) -> None:
        """Add observations and refresh the models (ask_tell_optimization.py:642-729)."""
        new_data = _as_map(new_data)
        if self._datasets.keys() != new_data.keys():
            raise ValueError(f"new_data keys {new_data.keys()} doesn't match dataset keys {self._datasets.keys()}")
        for tag, ds in new_data.items():
            self._datasets[tag] = (self._datasets[tag] + ds) if self._track_data else ds
        self._filter_datasets()
        # `fit_model` governs the constructor's initial fit only: a pre-trained model handed over with
        # fit_model=False (the reference's from_record / from_state, :497) still has to see the new observations
        # (ask_tell_optimization.py:716-718)
        for tag, model in self._models.items():
            self.update_model(model, self._datasets[tag])


class AskTellOptimizerNoTraining(AskTellOptimizer):
    """The Ask-Tell loop for models the caller trains and updates (ask_tell_optimization.py:749-757): neither the
    constructor nor ``tell`` touches the models."""

    def update_model(self, model, dataset: Dataset) -> None:
        pass
