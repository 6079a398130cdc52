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
