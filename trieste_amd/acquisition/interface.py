"""Acquisition-function plug-in surfaces (reference trieste/acquisition/interface.py:27-157)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, Mapping, Optional

from ..data import Dataset

AcquisitionFunction = Callable
"""x [..., B, D] -> [..., 1] (reference interface.py:27-38)."""


class AcquisitionFunctionClass(ABC):
    """An acquisition function written as a class with mutable state (interface.py:41-49)."""

    @abstractmethod
    def __call__(self, x):
        ...


class AcquisitionFunctionBuilder(ABC):
    """Builds an acquisition function from tagged models and datasets."""

    @abstractmethod
    def prepare_acquisition_function(self, models: Mapping, datasets: Optional[Mapping] = None):
        ...

    def update_acquisition_function(self, function, models: Mapping, datasets: Optional[Mapping] = None):
        return self.prepare_acquisition_function(models, datasets=datasets)


class SingleModelAcquisitionBuilder(ABC):
    """Builder needing one (model, dataset) pair; ``using(tag)`` adapts it (interface.py:96-131)."""

    def using(self, tag) -> AcquisitionFunctionBuilder:
        single = self

        class _Anon(AcquisitionFunctionBuilder):
            def __init__(self):
                self.single_builder = single

            def prepare_acquisition_function(self, models, datasets=None):
                return single.prepare_acquisition_function(
                    models[tag], dataset=None if datasets is None else datasets[tag])

            def update_acquisition_function(self, function, models, datasets=None):
                return single.update_acquisition_function(
                    function, models[tag], dataset=None if datasets is None else datasets[tag])

            def __repr__(self) -> str:
                return f"{single!r} using tag {tag!r}"

        return _Anon()

    @abstractmethod
    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None):
        ...

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None):
        return self.prepare_acquisition_function(model, dataset=dataset)


class GreedyAcquisitionFunctionBuilder(ABC):
    """Builds an acquisition function used to pick a batch greedily: ``pending_points`` are the
    points already chosen for the current batch (interface.py:160-216)."""

    @abstractmethod
    def prepare_acquisition_function(self, models: Mapping, datasets: Optional[Mapping] = None,
                                     pending_points=None):
        ...

    def update_acquisition_function(self, function, models: Mapping, datasets: Optional[Mapping] = None,
                                    pending_points=None, new_optimization_step: bool = True):
        return self.prepare_acquisition_function(models, datasets=datasets, pending_points=pending_points)


class SingleModelGreedyAcquisitionBuilder(ABC):
    """Single-model convenience form of :class:`GreedyAcquisitionFunctionBuilder` (interface.py:219-309)."""

    def using(self, tag) -> GreedyAcquisitionFunctionBuilder:
        single = self

        class _Anon(GreedyAcquisitionFunctionBuilder):
            def __init__(self):
                self.single_builder = single

            def prepare_acquisition_function(self, models, datasets=None, pending_points=None):
                return single.prepare_acquisition_function(
                    models[tag], dataset=None if datasets is None else datasets[tag], pending_points=pending_points)

            def update_acquisition_function(self, function, models, datasets=None, pending_points=None,
                                            new_optimization_step=True):
                return single.update_acquisition_function(
                    function, models[tag], dataset=None if datasets is None else datasets[tag],
                    pending_points=pending_points, new_optimization_step=new_optimization_step)

            def __repr__(self) -> str:
                return f"{single!r} using tag {tag!r}"

        return _Anon()

    @abstractmethod
    def prepare_acquisition_function(self, model, dataset: Optional[Dataset] = None, pending_points=None):
        ...

    def update_acquisition_function(self, function, model, dataset: Optional[Dataset] = None, pending_points=None,
                                    new_optimization_step: bool = True):
        return self.prepare_acquisition_function(model, dataset=dataset, pending_points=pending_points)


class VectorizedAcquisitionFunctionBuilder(AcquisitionFunctionBuilder):
    """Builds functions x [N, V, D] -> [N, V] whose V columns are optimised independently with
    :func:`batchify_vectorize` (interface.py:312-318)."""


class SingleModelVectorizedAcquisitionBuilder(SingleModelAcquisitionBuilder):
    """Single-model convenience form of :class:`VectorizedAcquisitionFunctionBuilder` (interface.py:321-376)."""

    def using(self, tag) -> AcquisitionFunctionBuilder:
        single = self

        class _Anon(VectorizedAcquisitionFunctionBuilder):
            def __init__(self):
                self.single_builder = single

            def prepare_acquisition_function(self, models, datasets=None):
                return single.prepare_acquisition_function(
                    models[tag], dataset=None if datasets is None else datasets[tag])

            def update_acquisition_function(self, function, models, datasets=None):
                return single.update_acquisition_function(
                    function, models[tag], dataset=None if datasets is None else datasets[tag])

            def __repr__(self) -> str:
                return f"{single!r} using tag {tag!r}"

        return _Anon()
