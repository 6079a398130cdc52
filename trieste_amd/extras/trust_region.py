"""Trust-region rules for box spaces over GLOBAL models and datasets (reference trieste/acquisition/rule.py:
UpdatableTrustRegion 1039-1237, BatchTrustRegionState 1240-1258, BatchTrustRegion 1261-1566, HypercubeTrustRegion
1569-1777, UpdatableTrustRegionBox 1780-1820, SingleObjectiveTrustRegionBox 1823-1860, BatchTrustRegionBox 1863-1920,
TREGOBox 1923-2035, TURBOBox 2038-2218).

These rules are host orchestration over the same posterior arithmetic: each region is a ``Box`` whose bounds move
with the data, and the base rule (EGO by default) maximises its acquisition function inside it -- every sweep,
top-k and L-BFGS-B refinement runs on the engine exactly as for the global space.  What is restated is the
single-objective region logic (success test with ``kappa`` x volume, growth / shrinkage by ``beta``, re-initialisation
below ``min_eps``, TREGO's global / local alternation) and the rule plumbing (``acquire`` / ``filter_datasets`` state
functions), and TURBO's box.  Local models / local datasets per region, product and discrete regions are not built: with
global models the base rule runs once per region, on a copy of the rule (the reference batches EGO over a tagged
multi-space instead; for one region -- TREGO -- and for vectorized builders the two coincide).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass
from typing import Mapping, Optional, Sequence, Tuple

import numpy as np

from ..data import OBJECTIVE
from ..space import Box, SearchSpace
from ..acquisition.rule import AcquisitionRule, EfficientGlobalOptimization


class UpdatableTrustRegionBox(Box):
    """A box with a centre (``location``) inside a global box (rule.py:1780-1820)."""

    def __init__(self, global_search_space: Box, region_index: Optional[int] = None):
        if not isinstance(global_search_space, Box):
            raise TypeError(f"search space should be a Box, got {type(global_search_space)}")
        super().__init__(global_search_space.lower, global_search_space.upper)
        self._global_search_space = global_search_space
        self.region_index = region_index
        self._initialized = False
        self._location = np.asarray(global_search_space.sample(1), dtype=np.float64)[0]

    @property
    def location(self) -> np.ndarray:
        return self._location

    @location.setter
    def location(self, location) -> None:
        self._location = np.asarray(location, dtype=np.float64).reshape(-1)

    @property
    def global_search_space(self) -> Box:
        return self._global_search_space

    def _get_bounds_within_distance(self, eps) -> Tuple[np.ndarray, np.ndarray]:
        return (np.maximum(self._global_search_space.lower, self.location - eps),
                np.minimum(self._global_search_space.upper, self.location + eps))

    def _set_bounds(self, lower, upper) -> None:
        self._lower = np.asarray(lower, dtype=np.float64)
        self._upper = np.asarray(upper, dtype=np.float64)

    def contains(self, points) -> np.ndarray:
        """Row-wise membership of points [N, D] -> [N] bool."""
        p = np.asarray(points, dtype=np.float64)
        return np.all((p >= self.lower) & (p <= self.upper), axis=-1)


class SingleObjectiveTrustRegionBox(UpdatableTrustRegionBox):
    """A hypercube region updated from the best observation inside it (HypercubeTrustRegion, rule.py:1569-1777 +
    SingleObjectiveTrustRegionBox, 1823-1860): a step is a success if the region's minimum improved by more than
    ``kappa`` x region volume; the size grows by 1 / ``beta`` after a success and shrinks by ``beta`` otherwise;
    below ``min_eps`` the region is re-initialised at a fresh location."""

    def __init__(self, global_search_space: Box, beta: float = 0.7, kappa: float = 1e-4, zeta: float = 0.5,
                 min_eps: float = 1e-2, region_index: Optional[int] = None):
        super().__init__(global_search_space, region_index)
        self._beta, self._kappa, self._zeta, self._min_eps = beta, kappa, zeta, min_eps
        self._step_is_success = False
        self._init_eps()
        self._update_domain()
        self._y_min = np.inf  # nothing observed yet: the first step always succeeds

    # -- size ----------------------------------------------------------------------------------------
    @property
    def eps(self) -> np.ndarray:
        return self._eps

    @eps.setter
    def eps(self, eps) -> None:
        self._eps = np.asarray(eps, dtype=np.float64)

    def _init_eps(self) -> None:
        self.eps = self._zeta * (self._global_search_space.upper - self._global_search_space.lower)

    def _update_domain(self) -> None:
        self._set_bounds(*self._get_bounds_within_distance(self.eps))

    @property
    def requires_initialization(self) -> bool:
        return (not self._initialized) or bool(np.any(self.eps < self._min_eps))

    # -- life cycle -----------------------------------------------------------------------------------
    def initialize(self, models: Optional[Mapping] = None, datasets: Optional[Mapping] = None,
                   location_candidate=None) -> None:
        """A fresh location (sampled from the global space unless given), full size, no history (:1641-1668)."""
        self.location = (location_candidate if location_candidate is not None
                         else np.asarray(self._global_search_space.sample(1))[0])
        self._step_is_success = False
        self._init_eps()
        self._update_domain()
        self._y_min = np.inf
        self._initialized = True

    def update(self, models: Optional[Mapping] = None, datasets: Optional[Mapping] = None) -> None:
        """Move / resize from the latest data (:1670-1709)."""
        x_min, y_min = self.get_dataset_min(datasets)
        tr_volume = float(np.prod(self.upper - self.lower))
        self._step_is_success = bool(y_min < self._y_min - self._kappa * tr_volume)
        self.eps = self.eps / self._beta if self._step_is_success else self.eps * self._beta
        if self._step_is_success:  # the centre only follows successful steps
            self.location = x_min
            self._y_min = float(y_min)
        self._update_domain()

    def get_values_min(self, query_points, values, num_query_points: Optional[int] = None,
                       in_region_only: bool = True) -> Tuple[np.ndarray, float]:
        """(argmin point [D], min value) of values [N, 1] over the query points, optionally only those inside the
        region and only the last ``num_query_points`` (:1711-1754); +inf if nothing qualifies."""
        qps = np.asarray(query_points, dtype=np.float64)
        vals = np.asarray(values, dtype=np.float64).reshape(len(qps), -1)[:, 0]
        if num_query_points is not None:
            qps, vals = qps[-num_query_points:], vals[-num_query_points:]
        if in_region_only:
            vals = np.where(self.contains(qps), vals, np.inf)
        ix = int(np.argmin(vals))
        return qps[ix], float(vals[ix])

    def get_dataset_min(self, datasets: Optional[Mapping]) -> Tuple[np.ndarray, float]:
        if datasets is None or len(datasets) != 1 or next(iter(datasets)) != OBJECTIVE:
            raise ValueError("a single OBJECTIVE dataset must be provided")
        dataset = next(iter(datasets.values()))
        return self.get_values_min(dataset.query_points, dataset.observations, in_region_only=True)


class TREGOBox(SingleObjectiveTrustRegionBox):
    """TREGO (Diouane et al. 2022; rule.py:1923-2035): alternate regular EGO steps over the global space with local
    steps inside the trust region.  Starts global; after a successful step the next one is global, after an
    unsuccessful one the mode flips; the size only changes in local mode; the centre is the global best point."""

    def __init__(self, global_search_space: Box, beta: float = 0.7, kappa: float = 1e-4, zeta: float = 0.5,
                 min_eps: float = 1e-2, region_index: Optional[int] = None):
        self._is_global = False
        super().__init__(global_search_space, beta, kappa, zeta, min_eps, region_index)

    @property
    def eps(self) -> np.ndarray:
        return self._eps

    @eps.setter
    def eps(self, eps) -> None:
        if not self._is_global:  # the size is frozen in global mode
            self._eps = np.asarray(eps, dtype=np.float64)

    def _update_domain(self) -> None:
        self._is_global = self._step_is_success or not self._is_global
        if self._is_global:
            self._set_bounds(self._global_search_space.lower, self._global_search_space.upper)
        else:
            super()._update_domain()

    def initialize(self, models=None, datasets=None, location_candidate=None) -> None:
        # global mode at construction, local mode for re-initialisations (_update_domain flips the flag)
        self._is_global = self._initialized
        super().initialize(models, datasets, location_candidate=location_candidate)

    def get_dataset_min(self, datasets: Optional[Mapping]) -> Tuple[np.ndarray, float]:
        if datasets is None or len(datasets) != 1 or next(iter(datasets)) != OBJECTIVE:
            raise ValueError("a single OBJECTIVE dataset must be provided")
        dataset = next(iter(datasets.values()))
        return self.get_values_min(dataset.query_points, dataset.observations, in_region_only=False)  # global minimum


class TURBOBox(UpdatableTrustRegionBox):
    """TURBO's region (Eriksson et al. 2019; rule.py:2038-2218): a box centred on the best observation whose side
    lengths follow the model's lengthscales at fixed volume ``L^D``; ``L`` doubles after ``success_tolerance``
    consecutive improvements, halves after ``failure_tolerance`` consecutive failures, is capped at ``L_max`` and
    restarts from ``L_init`` below ``L_min``.  Here it works with the global model (the reference pairs it with a
    local model per region)."""

    def __init__(self, global_search_space: Box, L_min: Optional[float] = None, L_init: Optional[float] = None,
                 L_max: Optional[float] = None, success_tolerance: int = 3, failure_tolerance: Optional[int] = None,
                 region_index: Optional[int] = None):
        super().__init__(global_search_space, region_index)
        width = float(np.max(global_search_space.upper - global_search_space.lower))
        L_min = (0.5 ** 7) * width if L_min is None else L_min
        L_init = 0.8 * width if L_init is None else L_init
        L_max = 1.6 * width if L_max is None else L_max
        for name, value in (("L_min", L_min), ("L_init", L_init), ("L_max", L_max)):
            if value <= 0:
                raise ValueError(f"{name} must be postive, got {value}")
        self.L_min, self.L_init, self.L_max = L_min, L_init, L_max
        self.L = L_init
        self.success_tolerance = success_tolerance
        self.failure_tolerance = failure_tolerance if failure_tolerance is not None else global_search_space.dimension
        self.success_counter = 0
        self.failure_counter = 0
        if self.success_tolerance <= 0:
            raise ValueError(f"success tolerance must be an integer greater than 0, got {self.success_tolerance}")
        if self.failure_tolerance <= 0:
            raise ValueError(f"success tolerance must be an integer greater than 0, got {self.failure_tolerance}")
        self.y_min = np.inf
        self.tr_width = global_search_space.upper - global_search_space.lower  # the full space to begin with
        self._update_domain()

    @property
    def requires_initialization(self) -> bool:
        return not self._initialized

    def _set_tr_width(self, models: Optional[Mapping]) -> None:
        """Stretch the region along the model's lengthscales at fixed volume (rule.py:2113-2138)."""
        if models is None or len(models) != 1 or next(iter(models)) != OBJECTIVE:
            raise ValueError("a single OBJECTIVE model must be provided")
        model = next(iter(models.values()))
        if not hasattr(model, "get_kernel"):
            raise TypeError(f"the model should support get_kernel, got {type(model)}")
        d = self._global_search_space.dimension
        lengthscales = np.broadcast_to(np.asarray(model.get_kernel().lengthscales, dtype=np.float64), (d,))
        self.tr_width = lengthscales * self.L / np.prod(lengthscales) ** (1.0 / d)

    def _update_domain(self) -> None:
        self._set_bounds(np.maximum(self._global_search_space.lower, self.location - self.tr_width / 2.0),
                         np.minimum(self._global_search_space.upper, self.location + self.tr_width / 2.0))

    def initialize(self, models: Optional[Mapping] = None, datasets: Optional[Mapping] = None,
                   location_candidate=None) -> None:
        x_min, self.y_min = self.get_dataset_min(datasets)
        self.location = x_min
        self.L, self.failure_counter, self.success_counter = self.L_init, 0, 0
        self._set_tr_width(models)
        self._update_domain()
        self._initialized = True

    def update(self, models: Optional[Mapping] = None, datasets: Optional[Mapping] = None) -> None:
        x_min, y_min = self.get_dataset_min(datasets)
        self.location = x_min
        step_is_success = y_min < self.y_min - 1e-10
        self.y_min = y_min
        self.failure_counter = 0 if step_is_success else self.failure_counter + 1
        self.success_counter = self.success_counter + 1 if step_is_success else 0
        if self.success_counter == self.success_tolerance:
            self.L *= 2.0
            self.success_counter = 0
        elif self.failure_counter == self.failure_tolerance:
            self.L *= 0.5
            self.failure_counter = 0
        self.L = min(self.L, self.L_max)
        if self.L < self.L_min:  # too small: start again
            self.L, self.failure_counter, self.success_counter = self.L_init, 0, 0
        self._set_tr_width(models)
        self._update_domain()

    def get_dataset_min(self, datasets: Optional[Mapping]) -> Tuple[np.ndarray, float]:
        """The best observation of the whole data set (rule.py:2202-2218)."""
        if datasets is None or len(datasets) != 1 or next(iter(datasets)) != OBJECTIVE:
            raise ValueError("a single OBJECTIVE dataset must be provided")
        dataset = next(iter(datasets.values()))
        ix = int(np.argmin(np.asarray(dataset.observations)[:, 0]))
        return np.asarray(dataset.query_points)[ix], float(np.asarray(dataset.observations)[ix, 0])


@dataclass(frozen=True)
class BatchTrustRegionState:
    """The acquisition state of :class:`BatchTrustRegionBox`: its regions (rule.py:1240-1258)."""

    subspaces: Sequence[UpdatableTrustRegionBox]
    subspace_tags: Sequence[str]

    def __deepcopy__(self, memo):
        return BatchTrustRegionState(copy.deepcopy(tuple(self.subspaces), memo), tuple(self.subspace_tags))


def get_unique_points_mask(points, tolerance: float = 1e-6) -> np.ndarray:
    """mask[i] is False iff point i is within ``tolerance`` of an earlier kept point (reference utils)."""
    pts = np.asarray(points, dtype=np.float64)
    mask = np.ones(len(pts), dtype=bool)
    for i in range(len(pts)):
        for j in range(i):
            if mask[j] and np.linalg.norm(pts[i] - pts[j]) <= tolerance:
                mask[i] = False
                break
    return mask


class BatchTrustRegionBox(AcquisitionRule):
    """Query points per trust region, each region with its own base-rule instance (rule.py:1261-1566, 1863-1920).
    ``acquire`` returns ``state -> (state, points [N, V, D])`` (N points of the base rule in each of the V regions); ``filter_datasets`` returns
    ``state -> (state, datasets)`` and is where the regions are updated from the newest data (the loops call it
    after every observation).  Regions whose centres coincide are re-initialised."""

    def __init__(self, init_subspaces=None, rule: Optional[AcquisitionRule] = None):
        self._init_subspaces = None
        self._tags = None
        if init_subspaces is not None:
            if not isinstance(init_subspaces, Sequence):
                init_subspaces = [init_subspaces]
            self._init_subspaces = tuple(init_subspaces)
            for index, subspace in enumerate(self._init_subspaces):
                subspace.region_index = index
            self._tags = tuple(str(i) for i in range(len(self._init_subspaces)))
        self._rule = rule
        self._rules = None

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self._init_subspaces!r}, {self._rule!r})"

    @property
    def num_local_datasets(self) -> int:
        if self._init_subspaces is None:
            raise ValueError("the subspaces have not been initialized")
        return len(self._init_subspaces)

    def initialize_subspaces(self, search_space: SearchSpace) -> None:
        """Default: one :class:`SingleObjectiveTrustRegionBox` over the global space (rule.py:1869-1890)."""
        if self._init_subspaces is None:
            if not isinstance(search_space, Box):
                raise TypeError(f"search space should be a Box, got {type(search_space)}")
            self._init_subspaces = (SingleObjectiveTrustRegionBox(search_space, region_index=0),)
            self._tags = ("0",)

    def _region_rules(self, count: int):
        if self._rule is None:  # the reference's defaults (rule.py:1351-1360)
            if isinstance(self._init_subspaces[0], TURBOBox):
                from ..acquisition.rule import DiscreteThompsonSampling

                dim = self._init_subspaces[0].global_search_space.dimension
                self._rule = DiscreteThompsonSampling(min(100 * dim, 5_000), 1)
            else:
                self._rule = EfficientGlobalOptimization()
        if self._rules is None:
            base = self._rule
            q = getattr(base, "_num_query_points", 1)
            if isinstance(base, EfficientGlobalOptimization) and q > 1:
                # the reference optimises such a rule ONCE over the tagged product of the regions, column v of the
                # batch inside region v (rule.py:1476-1493).  For a vectorized builder the columns are independent
                # functions, so one single-point rule per region is the same computation
                from ..acquisition.interface import VectorizedAcquisitionFunctionBuilder

                if q != count:
                    raise ValueError(f"the base rule asks for {q} query points but there are {count} trust regions")
                if not isinstance(base._builder, VectorizedAcquisitionFunctionBuilder):
                    raise NotImplementedError("joint / greedy batch builders over several trust regions need the "
                                              "tagged multi-space optimisation of the reference, which is not built")
                if base._acquisition_function is not None:
                    raise ValueError("the base rule must not have been used before")
                inner = getattr(base._builder, "single_builder", base._builder)
                self._rules = [EfficientGlobalOptimization(copy.deepcopy(inner), base._base_optimizer, 1)
                               for _ in range(count)]
            else:  # any other rule runs per region as it is: its N points x V regions make the batch
                self._rules = [copy.deepcopy(base) for _ in range(count)]
        return self._rules

    def _subspaces_of(self, state: Optional[BatchTrustRegionState]):
        if state is None:
            return self._init_subspaces
        if tuple(state.subspace_tags) != self._tags:
            raise ValueError(f"The tags of the state acquisition space {state.subspace_tags} should be the same as "
                             f"the tags of the BatchTrustRegion acquisition rule {self._tags}")
        return tuple(state.subspaces)

    def acquire(self, search_space: SearchSpace, models: Mapping, datasets: Optional[Mapping] = None):
        self.initialize_subspaces(search_space)
        for subspace in self._init_subspaces:
            gs = subspace.global_search_space
            if not (np.array_equal(gs.lower, search_space.lower) and np.array_equal(gs.upper, search_space.upper)):
                raise ValueError("The global search space of the subspaces should be the same as the search space "
                                 "passed to the BatchTrustRegionBox acquisition rule.")
        rules = self._region_rules(len(self._init_subspaces))

        def state_func(state: Optional[BatchTrustRegionState]):
            subspaces = self._subspaces_of(state)
            points = [np.asarray(rule.acquire(subspace, models, datasets)) for subspace, rule in zip(subspaces, rules)]
            stacked = np.stack(points, axis=1)  # [N, V, D]
            return BatchTrustRegionState(subspaces, self._tags), stacked.reshape(-1, len(subspaces), stacked.shape[-1])

        return state_func

    def get_initialize_subspaces_mask(self, subspaces, models, datasets=None) -> np.ndarray:
        """Regions with a non-unique centre start afresh (rule.py:1912-1920)."""
        return ~get_unique_points_mask(np.stack([s.location for s in subspaces]), tolerance=1e-6)

    def filter_datasets(self, models: Mapping, datasets: Mapping):
        if self._init_subspaces is None:
            raise ValueError("the subspaces have not been initialized")

        def state_func(state: Optional[BatchTrustRegionState]):
            subspaces = copy.deepcopy(self._subspaces_of(state))  # never modify the caller's copies
            for subspace in subspaces:
                if subspace.requires_initialization:
                    subspace.initialize(models, datasets)
                else:
                    subspace.update(models, datasets)
            for subspace, again in zip(subspaces, self.get_initialize_subspaces_mask(subspaces, models, datasets)):
                if again:
                    subspace.initialize(models, datasets)
            return BatchTrustRegionState(subspaces, self._tags), datasets  # global datasets pass unfiltered

        return state_func
