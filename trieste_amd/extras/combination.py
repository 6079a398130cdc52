"""Builders whose acquisition function combines the outputs of other builders' functions (reference
trieste/acquisition/combination.py: Reducer 28-119, Sum 122-136, Product 139-153, Map 156-186).  Each constituent
function is evaluated by its own model's engine; the reduction is elementwise arithmetic on the returned arrays
(several models / engines are involved, so there is no fused arg-max: optimizers take their generic path)."""
from __future__ import annotations

from abc import abstractmethod
from typing import Callable, Mapping, Optional, Sequence

import numpy as np

from ..acquisition.interface import AcquisitionFunctionBuilder


class Reducer(AcquisitionFunctionBuilder):
    r"""Builds a function whose output is computed by :meth:`_reduce` from the outputs of the functions of the
    given builders."""

    def __init__(self, *builders: AcquisitionFunctionBuilder):
        if len(builders) == 0:
            raise ValueError("At least one acquisition builder expected, got none.")
        self._acquisitions = builders
        self.functions: tuple = ()

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({', '.join(map(repr, self._acquisitions))})"

    @property
    def acquisitions(self) -> Sequence[AcquisitionFunctionBuilder]:
        return self._acquisitions

    def _function(self):
        def evaluate_acquisition_function_fn(at):
            return self._reduce_acquisition_functions(at, self.functions)

        return evaluate_acquisition_function_fn

    def prepare_acquisition_function(self, models: Mapping, datasets: Optional[Mapping] = None):
        self.functions = tuple(acq.prepare_acquisition_function(models, datasets=datasets) for acq in self.acquisitions)
        return self._function()

    def update_acquisition_function(self, function, models: Mapping, datasets: Optional[Mapping] = None):
        self.functions = tuple(acq.update_acquisition_function(fn, models, datasets=datasets)
                               for fn, acq in zip(self.functions, self.acquisitions))
        return self._function()

    def _reduce_acquisition_functions(self, at, acquisition_functions):
        return self._reduce([np.asarray(fn(at), dtype=np.float64) for fn in acquisition_functions])

    @abstractmethod
    def _reduce(self, inputs):
        ...


class Sum(Reducer):
    """Element-wise sum of the constituent functions' outputs."""

    def _reduce(self, inputs):
        return np.sum(np.stack(inputs, axis=0), axis=0)


class Product(Reducer):
    """Element-wise product of the constituent functions' outputs."""

    def _reduce(self, inputs):
        return np.prod(np.stack(inputs, axis=0), axis=0)


class Map(Reducer):
    """Applies ``map_fn`` to the output of a single builder's function (``Map(lambda x: -x, builder)``)."""

    def __init__(self, map_fn: Callable, builder: AcquisitionFunctionBuilder):
        super().__init__(builder)
        self._map_fn = map_fn

    def _reduce(self, inputs):
        if len(inputs) != 1:
            raise ValueError(f"Map expects exactly one input, got {len(inputs)}")
        return self._map_fn(inputs[0])
