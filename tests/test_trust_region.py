"""CPU tests of the trust-region rules (reference tests/unit/acquisition/test_rule.py:595-870 for TREGO and the
single-objective box region), the engine replaced at its boundary by tests/fakes.py::FakeEngine."""
import copy

import numpy as np
import pytest

import trieste_amd.models as M
from tests.fakes import FakeEngine
from trieste_amd import objectives as OBJ
from trieste_amd.extras import (BatchTrustRegionBox, BatchTrustRegionState, DiscreteThompsonSampling,
                                     EfficientGlobalOptimization, SingleObjectiveTrustRegionBox, TREGOBox, TURBOBox,
                                     generate_continuous_optimizer)
from trieste_amd.acquisition.rule import AcquisitionRule
from trieste_amd.ask_tell_optimization import AskTellOptimizer, AskTellOptimizerNoTraining
from trieste_amd.bayesian_optimizer import BayesianOptimizer
from trieste_amd.data import OBJECTIVE, Dataset
from trieste_amd.space import Box


@pytest.fixture(autouse=True)
def fake_engine(monkeypatch):
    monkeypatch.setattr(M, "GPEngine", FakeEngine)


LOWER, UPPER = np.array([-2.2, -1.0]), np.array([1.3, 3.3])


class _Midpoint(AcquisitionRule):
    """The centre of whatever space it is given (reference test_rule.py:585-595)."""

    def acquire(self, search_space, models, datasets=None):
        return ((search_space.upper + search_space.lower) / 2).reshape(1, -1)


def _subspace(search_space, acquisition_space, dataset, eps, previous_y_min, is_global):
    """reference trego_create_subspace (test_rule.py:630-647)."""
    subspace = TREGOBox(search_space, region_index=0)
    subspace.initialize(datasets={OBJECTIVE: dataset})
    subspace._eps = np.asarray(eps, dtype=float)
    subspace._y_min = float(previous_y_min)
    subspace._is_global = is_global
    subspace._set_bounds(acquisition_space.lower, acquisition_space.upper)
    subspace.location = (acquisition_space.lower + acquisition_space.upper) / 2
    return subspace


def _step(subspace, dataset, rule=None):
    space = Box(LOWER, UPPER)
    tr = BatchTrustRegionBox(subspace, rule or _Midpoint())
    state0 = BatchTrustRegionState([subspace], ["0"])
    state, point = tr.acquire(space, {OBJECTIVE: None}, datasets={OBJECTIVE: dataset})(state0)
    state, filtered = tr.filter_datasets({OBJECTIVE: None}, {OBJECTIVE: dataset})(state)
    return state.subspaces[0], point, filtered


def test_trego_for_default_state():
    space = Box(LOWER, UPPER)  # :604-628
    dataset = Dataset(np.array([[0.1, 0.2]]), np.array([[0.012]]))
    tr = BatchTrustRegionBox(TREGOBox(space), _Midpoint())
    state, point = tr.acquire_single(space, None, dataset=dataset)(None)
    assert point.shape == (1, 1, 2)
    np.testing.assert_allclose(point[0, 0], [-0.45, 1.15])
    state, _ = tr.filter_datasets({OBJECTIVE: None}, {OBJECTIVE: dataset})(state)
    sub = state.subspaces[0]
    assert isinstance(sub, TREGOBox) and sub._is_global and sub._y_min == np.inf
    np.testing.assert_allclose(sub.lower, LOWER)
    np.testing.assert_allclose(sub.upper, UPPER)


def test_trego_successful_global_to_global_trust_region_unchanged():
    space = Box(LOWER, UPPER)  # :660-697
    dataset = Dataset(np.array([[0.1, 0.2], [-0.1, -0.2]]), np.array([[0.4], [0.3]]))
    eps = 0.5 * (UPPER - LOWER) / 10
    sub, point, _ = _step(_subspace(space, space, dataset, eps, 0.4, True), dataset)
    np.testing.assert_allclose(sub._eps, eps)
    assert sub._is_global
    np.testing.assert_allclose(point[0, 0], [-0.45, 1.15])
    np.testing.assert_allclose(sub.lower, LOWER)
    np.testing.assert_allclose(sub.upper, UPPER)
    np.testing.assert_allclose(sub.location, [-0.1, -0.2])  # the centre follows the success


def test_trego_unsuccessful_global_to_local_trust_region_unchanged():
    space = Box(LOWER, UPPER)  # :707-744
    dataset = Dataset(np.array([[0.1, 0.2], [-0.1, -0.2]]), np.array([[0.4], [0.5]]))
    eps = 0.5 * (UPPER - LOWER) / 10
    before = _subspace(space, space, dataset, eps, 0.4, True)
    sub, point, _ = _step(copy.deepcopy(before), dataset)
    np.testing.assert_allclose(sub._eps, eps)
    assert not sub._is_global
    assert np.all(LOWER < sub.lower) and np.all(sub.upper < UPPER)
    assert point[0, 0] in before


def test_trego_successful_local_to_global_trust_region_increased():
    space = Box(LOWER, UPPER)  # :754-781
    dataset = Dataset(np.array([[0.1, 0.2], [-0.1, -0.2]]), np.array([[0.4], [0.3]]))
    eps = 0.5 * (UPPER - LOWER) / 10
    local = Box(dataset.query_points[0] - eps, dataset.query_points[0] + eps)
    sub, _, _ = _step(_subspace(space, local, dataset, eps, 0.4, False), dataset)
    assert np.all(eps < sub._eps) and sub._is_global
    np.testing.assert_allclose(sub.lower, LOWER)
    np.testing.assert_allclose(sub.upper, UPPER)


def test_trego_unsuccessful_local_to_global_trust_region_reduced():
    space = Box(LOWER, UPPER)  # :791-818
    dataset = Dataset(np.array([[0.1, 0.2], [-0.1, -0.2]]), np.array([[0.4], [0.5]]))
    eps = 0.5 * (UPPER - LOWER) / 10
    local = Box(dataset.query_points[0] - eps, dataset.query_points[0] + eps)
    sub, _, _ = _step(_subspace(space, local, dataset, eps, 0.4, False), dataset)
    assert np.all(sub._eps < eps) and sub._is_global
    np.testing.assert_allclose(sub.lower, LOWER)
    np.testing.assert_allclose(sub.upper, UPPER)


def test_trego_always_uses_global_dataset_and_state_is_copied():
    space = Box([0.0, 0.0], [1.0, 1.0])  # :821-857
    dataset = Dataset(np.array([[0.1, 0.2], [-0.1, -0.2], [1.1, 2.3]]), np.array([[0.4], [0.5], [0.6]]))
    tr = BatchTrustRegionBox(TREGOBox(space), _Midpoint())
    state, _ = tr.acquire(space, {OBJECTIVE: None}, {OBJECTIVE: dataset})(None)
    more = dataset + Dataset(np.array([[0.5, -0.2], [0.7, 0.2], [1.1, 0.3], [0.5, 0.5]]),
                             np.array([[0.7], [0.8], [0.9], [1.0]]))
    new_state, filtered = tr.filter_datasets({OBJECTIVE: None}, {OBJECTIVE: more})(state)
    np.testing.assert_array_equal(filtered[OBJECTIVE].query_points, more.query_points)  # nothing is filtered out
    assert new_state.subspaces[0] is not state.subspaces[0]  # the caller's regions are never modified
    dc = copy.deepcopy(new_state)  # :845-870
    assert dc.subspaces[0] is not new_state.subspaces[0]
    np.testing.assert_array_equal(dc.subspaces[0].lower, new_state.subspaces[0].lower)


def test_single_objective_region_shrinks_grows_and_reinitialises():
    space = Box([0.0, 0.0], [1.0, 1.0])  # HypercubeTrustRegion.update / requires_initialization (rule.py:1632-1709)
    region = SingleObjectiveTrustRegionBox(space, beta=0.5, kappa=1e-4, zeta=0.5, min_eps=0.1)
    assert region.requires_initialization
    region.initialize(location_candidate=np.array([0.5, 0.5]))
    np.testing.assert_allclose(region.eps, [0.5, 0.5])
    np.testing.assert_allclose(region.lower, [0.0, 0.0])
    data = Dataset(np.array([[0.5, 0.5], [0.45, 0.55]]), np.array([[1.0], [0.5]]))
    region.update(datasets={OBJECTIVE: data})  # first step: always a success -> grows, moves to the best point
    np.testing.assert_allclose(region.eps, [1.0, 1.0])
    np.testing.assert_allclose(region.location, [0.45, 0.55])
    for expected in (0.5, 0.25, 0.125):  # no improvement: shrink by beta each time, centre stays
        region.update(datasets={OBJECTIVE: data})
        np.testing.assert_allclose(region.eps, [expected, expected])
        np.testing.assert_allclose(region.location, [0.45, 0.55])
        np.testing.assert_allclose(region.upper, np.minimum(1.0, region.location + expected))
    assert not region.requires_initialization
    region.update(datasets={OBJECTIVE: data})
    assert region.requires_initialization  # 0.0625 < min_eps
    # only points INSIDE the region count for its minimum
    outside = data + Dataset(np.array([[0.9, 0.9]]), np.array([[-10.0]]))
    _, y_min = region.get_dataset_min({OBJECTIVE: outside})
    assert y_min == 0.5
    with pytest.raises(ValueError):
        region.get_dataset_min({"foo": data})
    with pytest.raises(ValueError):
        region.get_dataset_min(None)


def test_batch_trust_region_box_rule_checks_and_duplicate_centres():
    space = Box([0.0, 0.0], [1.0, 1.0])
    data = Dataset(np.array([[0.2, 0.2], [0.8, 0.8]]), np.array([[1.0], [2.0]]))
    from trieste_amd.extras import BatchMonteCarloExpectedImprovement, ParallelContinuousThompsonSampling

    with pytest.raises(NotImplementedError):  # a joint batch builder across regions needs the tagged multi-space
        BatchTrustRegionBox([TREGOBox(space), TREGOBox(space)], EfficientGlobalOptimization(
            BatchMonteCarloExpectedImprovement(10), num_query_points=2)).acquire(space, {OBJECTIVE: None}, {OBJECTIVE: data})
    with pytest.raises(ValueError):  # as many batch elements as regions
        BatchTrustRegionBox([TREGOBox(space)], EfficientGlobalOptimization(
            ParallelContinuousThompsonSampling(), num_query_points=2)).acquire(space, {OBJECTIVE: None}, {OBJECTIVE: data})
    with pytest.raises(ValueError):  # a different global space
        BatchTrustRegionBox(TREGOBox(space), _Midpoint()).acquire(Box([0.0, 0.0], [2.0, 1.0]), {OBJECTIVE: None},
                                                                  {OBJECTIVE: data})
    rule = BatchTrustRegionBox(rule=_Midpoint())
    with pytest.raises(ValueError):
        rule.filter_datasets({OBJECTIVE: None}, {OBJECTIVE: data})
    state, pts = rule.acquire(space, {OBJECTIVE: None}, {OBJECTIVE: data})(None)  # default: one region
    assert pts.shape == (1, 1, 2) and rule.num_local_datasets == 1
    a, b = SingleObjectiveTrustRegionBox(space), SingleObjectiveTrustRegionBox(space)
    a.initialize(location_candidate=np.array([0.3, 0.3]))
    b.initialize(location_candidate=np.array([0.3, 0.3]))
    tr = BatchTrustRegionBox([a, b], _Midpoint())
    assert a.region_index == 0 and b.region_index == 1
    mask = tr.get_initialize_subspaces_mask([a, b], None, None)
    np.testing.assert_array_equal(mask, [False, True])  # the second of two coinciding regions starts afresh
    with pytest.raises(ValueError):  # a state from another rule
        tr.acquire(space, {OBJECTIVE: None}, {OBJECTIVE: data})(BatchTrustRegionState([a], ["7"]))


def _model(n=10, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, 2))
    data = Dataset(x, OBJ.scaled_branin(x))
    return M.GaussianProcessRegression(M.build_gpr(data, Box([0, 0], [1, 1]), likelihood_variance=1e-4)), data


def test_trego_through_the_loops_alternates_modes_and_improves():
    space = Box([0, 0], [1, 1])
    opt = generate_continuous_optimizer(num_initial_samples=300, num_optimization_runs=3)
    model, data = _model()
    rule = BatchTrustRegionBox(TREGOBox(space), EfficientGlobalOptimization(optimizer=opt))
    res = BayesianOptimizer(lambda x: Dataset(x, OBJ.scaled_branin(x)), space).optimize(8, data, model, rule, fit_model=False)
    final = res.final_result.unwrap()
    assert len(final.dataset) == 18
    assert final.dataset.observations.min() < data.observations.min()
    states = [r.acquisition_state for r in res.history[1:]] + [final.acquisition_state]
    modes = [s.subspaces[0]._is_global for s in states]
    assert all(isinstance(s, BatchTrustRegionState) for s in states) and True in modes and False in modes
    for s in states:  # a local step's region is a proper sub-box around the best point
        sub = s.subspaces[0]
        assert np.all(sub.lower >= 0) and np.all(sub.upper <= 1)
        if not sub._is_global:
            assert np.any(sub.upper - sub.lower < 1.0)
    # Ask-Tell: two regions, one point each
    model2, data2 = _model(seed=1)
    regions = [SingleObjectiveTrustRegionBox(space) for _ in range(2)]
    loop = AskTellOptimizerNoTraining(space, data2, model2,
                                      BatchTrustRegionBox(regions, EfficientGlobalOptimization(optimizer=opt)))
    for _ in range(3):
        pts = loop.ask()
        assert pts.shape == (1, 2, 2)
        for v, sub in enumerate(loop.acquisition_state.subspaces):
            assert pts[0, v] in sub
        flat = pts.reshape(-1, 2)
        loop.tell(Dataset(flat, OBJ.scaled_branin(flat)))
    # AskTellOptimizerNoTraining: the caller manages the model -- neither refitted nor updated
    # (reference ask_tell_optimization.py:749-757)
    assert len(loop.dataset) == 10 + 6 and model2.engine.N == 10


# ---- TURBOBox (reference test_rule.py:878-1246) -------------------------------------------------------------
class _KernelOnly:
    """A model exposing only what TURBOBox needs."""

    def __init__(self, lengthscales):
        self._k = M.SquaredExponential(1.0, np.asarray(lengthscales, dtype=float))

    def get_kernel(self):
        return self._k


def test_turbo_parameter_checks_and_heuristics():
    space = Box([-2.2, -1.0], [1.3, 3.3])
    for kwargs in ({"L_init": -1.0}, {"L_max": 0.0}, {"L_min": -0.1}, {"failure_tolerance": 0}, {"success_tolerance": -2}):
        with pytest.raises(ValueError):  # :897-915
            TURBOBox(space, **kwargs)
    big = Box([-2.0] * 20, [1.0] * 20)  # :918-938
    rule = BatchTrustRegionBox(TURBOBox(big))
    rule.acquire(big, {OBJECTIVE: _KernelOnly(np.ones(20))}, {OBJECTIVE: Dataset(np.zeros((1, 20)), np.zeros((1, 1)))})
    region = rule._init_subspaces[0]
    assert region.L_init == 0.8 * 3.0 and region.L_min == (0.5 ** 7) * 3.0 and region.L_max == 1.6 * 3.0
    assert region.failure_tolerance == 20
    assert isinstance(rule._rule, DiscreteThompsonSampling) and rule._rule._num_search_space_samples == 2_000
    rule = BatchTrustRegionBox(TURBOBox(big), rule=EfficientGlobalOptimization())
    rule.acquire(big, {OBJECTIVE: _KernelOnly(np.ones(20))}, None)
    assert isinstance(rule._rule, EfficientGlobalOptimization)
    with pytest.raises(ValueError):
        TURBOBox(space).get_dataset_min({"foo": Dataset(np.zeros((1, 2)), np.zeros((1, 1)))})
    with pytest.raises(ValueError):
        TURBOBox(space)._set_tr_width({"foo": _KernelOnly([1.0, 1.0])})


def _turbo_region(space, L, failure_counter, success_counter, previous_y_min):
    region = TURBOBox(space)  # reference turbo_create_region (test_rule.py:1000-1021)
    region.L, region.failure_counter, region.success_counter, region.y_min = L, failure_counter, success_counter, previous_y_min
    region._initialized = True
    return region


def _turbo_step(region, dataset, models):
    tr = BatchTrustRegionBox(region, _Midpoint())
    state = BatchTrustRegionState([region], ["0"])
    state, _ = tr.acquire(region.global_search_space, models, {OBJECTIVE: dataset})(state)
    state, _ = tr.filter_datasets(models, {OBJECTIVE: dataset})(state)
    return state.subspaces[0]


def test_turbo_changes_size_only_when_a_tolerance_is_reached_and_restarts_when_too_small():
    dataset = Dataset(np.array([[0.0, 0.0]]), np.array([[0.012]]))
    models = {OBJECTIVE: _KernelOnly([4.0, 1.0])}
    space = Box([0.0, 0.0], [1.0, 1.0])
    # success, below the tolerance: size unchanged, the box follows the lengthscales at fixed volume (:1024-1063)
    for fc in (0, 1):
        for sc in (0, 1):
            r = _turbo_step(_turbo_region(space, 0.8, fc, sc, 0.012 + 2.0), dataset, models)
            assert r.L == 0.8 and r.success_counter == sc + 1 and r.failure_counter == 0
            np.testing.assert_allclose(r.lower, [0.0, 0.0])
            np.testing.assert_allclose(r.upper, [0.8, 0.2])
    # failure, below the tolerance (:1065-1101)
    for sc in (0, 1, 2):
        r = _turbo_step(_turbo_region(space, 0.8, 0, sc, 0.012), dataset, models)
        assert r.L == 0.8 and r.success_counter == 0 and r.failure_counter == 1
    # the third success in a row doubles L (capped at L_max = 1.6), the second failure (D = 2) halves it (:1104-1177)
    r = _turbo_step(_turbo_region(space, 0.8, 0, 2, 0.012 + 2.0), dataset, models)
    assert r.L == 1.6 and r.success_counter == 0
    np.testing.assert_allclose(r.upper, [1.0, 0.4])
    r = _turbo_step(_turbo_region(space, 1.6, 0, 2, 0.012 + 2.0), dataset, models)
    assert r.L == 1.6  # capped
    r = _turbo_step(_turbo_region(space, 0.8, 1, 0, 0.012), dataset, models)
    assert r.L == 0.4 and r.failure_counter == 0
    np.testing.assert_allclose(r.upper, [0.4, 0.1])
    # below L_min the region restarts from L_init (:1180-1245)
    r = _turbo_step(_turbo_region(space, (0.5 ** 7) * 1.0, 1, 0, 0.012), dataset, models)
    assert r.L == 0.8 and r.failure_counter == 0 and r.success_counter == 0
    dc = copy.deepcopy(BatchTrustRegionState([r], ["0"]))
    assert dc.subspaces[0] is not r and dc.subspaces[0].L == r.L


def test_turbo_with_thompson_sampling_through_the_loop():
    space = Box([0, 0], [1, 1])
    model, data = _model(n=12, seed=2)
    res = BayesianOptimizer(lambda x: Dataset(x, OBJ.scaled_branin(x)), space).optimize(
        6, data, model, BatchTrustRegionBox(TURBOBox(space)), fit_model=False)
    final = res.final_result.unwrap()
    assert len(final.dataset) == 18 and final.dataset.observations.min() <= data.observations.min()
    region = final.acquisition_state.subspaces[0]
    best = final.dataset.query_points[int(np.argmin(final.dataset.observations[:, 0]))]
    np.testing.assert_allclose(region.location, best)  # centred on the best observation
    assert np.all(region.lower >= 0) and np.all(region.upper <= 1) and best in region
