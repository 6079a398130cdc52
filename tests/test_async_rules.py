"""CPU tests of the asynchronous rules and their state (reference tests/unit/acquisition/test_rule.py:492-571,
2629-2772) and of the loops' state threading (bayesian_optimizer.py:793-800, ask_tell_optimization.py:609-618).
The engine is replaced at its boundary by tests/fakes.py::FakeEngine (oracle-backed)."""
import numpy as np
import pytest

import trieste_amd.models as M
from tests.fakes import FakeEngine
from trieste_amd import objectives as OBJ
from trieste_amd.extras import (GIBBON, AsynchronousGreedy, AsynchronousOptimization, AsynchronousRuleState,
                                     BatchMonteCarloExpectedImprovement, Fantasizer, LocalPenalization,
                                     NegativeLowerConfidenceBound, generate_continuous_optimizer,
                                     generate_random_search_optimizer)
from trieste_amd.ask_tell_optimization import AskTellOptimizer
from trieste_amd.bayesian_optimizer import BayesianOptimizer
from trieste_amd.data import OBJECTIVE, Dataset
from trieste_amd.space import Box


@pytest.fixture(autouse=True)
def fake_engine(monkeypatch):
    monkeypatch.setattr(M, "GPEngine", FakeEngine)


SPACE = Box([0, 0], [1, 1])


def _model(n=12, noise=1e-2, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, 2))
    data = Dataset(x, OBJ.scaled_branin(x))
    return M.GaussianProcessRegression(M.build_gpr(data, SPACE, likelihood_variance=noise)), data


# ---- AsynchronousRuleState (test_rule.py:2629-2772) -------------------------------------------------------
def test_asynchronous_rule_state_pending_points_and_shapes():
    pending = np.array([[1.0], [2.0], [3.0]])
    np.testing.assert_array_equal(AsynchronousRuleState(pending).pending_points, pending)
    for bad in (np.array([1.0, 2.0]), np.array([[[1.0], [2.0]]])):
        with pytest.raises(ValueError):
            AsynchronousRuleState(bad)
    assert not AsynchronousRuleState(None).has_pending_points
    assert not AsynchronousRuleState(np.zeros((0, 2))).has_pending_points
    assert AsynchronousRuleState(pending).has_pending_points


def test_asynchronous_rule_state_remove_points():
    with pytest.raises(ValueError):
        AsynchronousRuleState(np.array([[1.0], [2.0], [3.0]])).remove_points(np.array([[1.0, 1.0]]))
    with pytest.raises(ValueError):
        AsynchronousRuleState(np.array([[1.0, 1.0], [2.0, 2.0]])).remove_points(np.array([[1.0]]))
    with pytest.raises(ValueError):
        AsynchronousRuleState(np.array([[1.0, 1.0], [2.0, 2.0]])).remove_points(np.array([[[1.0, 1.0], [2.0, 2.0]]]))
    p = np.array([[1.0], [2.0], [3.0]])
    cases = [(p, [[1.0]], [[2.0], [3.0]]), (p, [[2.0]], [[1.0], [3.0]]), (p, [[3.0]], [[1.0], [2.0]]),
             (p, [[4.0]], [[1.0], [2.0], [3.0]]),
             ([[1.0], [2.0], [3.0], [2.0]], [[2.0]], [[1.0], [3.0], [2.0]]),          # one occurrence only
             ([[1.0], [2.0], [3.0], [2.0]], [[2.0], [3.0]], [[1.0], [2.0]]),
             ([[1.0], [2.0], [3.0], [2.0]], [[2.0], [2.0]], [[1.0], [3.0]]),
             ([[1.0], [2.0], [3.0], [2.0]], [[2.0], [3.0], [4.0]], [[1.0], [2.0]]),
             ([[1.0, 1.0], [2.0, 3.0]], [[1.0, 1.0], [2.0, 2.0], [3.0, 3.0], [1.0, 2.0]], [[2.0, 3.0]])]
    for pending, remove, expected in cases:
        state = AsynchronousRuleState(np.array(pending)).remove_points(np.array(remove))
        np.testing.assert_array_equal(state.pending_points, expected)
    assert not AsynchronousRuleState(None).remove_points(np.array([[2.0]])).has_pending_points
    assert not AsynchronousRuleState(p).remove_points(p).has_pending_points


def test_asynchronous_rule_state_add_pending_points():
    with pytest.raises(ValueError):
        AsynchronousRuleState(np.array([[1.0], [2.0], [3.0]])).add_pending_points(np.array([[1.0, 1.0]]))
    with pytest.raises(ValueError):
        AsynchronousRuleState(np.array([[1.0, 1.0], [2.0, 2.0]])).add_pending_points(np.array([[1.0]]))
    with pytest.raises(ValueError):
        AsynchronousRuleState(np.array([[1.0, 1.0], [2.0, 2.0]])).add_pending_points(np.zeros((1, 2, 2)))
    np.testing.assert_array_equal(AsynchronousRuleState(None).add_pending_points(np.array([[1.0]])).pending_points, [[1.0]])
    np.testing.assert_array_equal(
        AsynchronousRuleState(np.array([[1.0], [2.0]])).add_pending_points(np.array([[1.0]])).pending_points,
        [[1.0], [2.0], [1.0]])
    np.testing.assert_array_equal(
        AsynchronousRuleState(np.array([[1.0, 1.0], [2.0, 2.0]])).add_pending_points(np.array([[3.0, 3.0], [4.0, 4.0]])).pending_points,
        [[1.0, 1.0], [2.0, 2.0], [3.0, 3.0], [4.0, 4.0]])


# ---- the rules (test_rule.py:492-571) ---------------------------------------------------------------------
def test_async_rules_argument_checks():
    with pytest.raises(NotImplementedError):
        AsynchronousGreedy(NegativeLowerConfidenceBound())
    with pytest.raises(ValueError):
        AsynchronousGreedy(None)
    for q in (0, -5):
        with pytest.raises(ValueError):
            AsynchronousOptimization(num_query_points=q)
        with pytest.raises(ValueError):
            AsynchronousGreedy(LocalPenalization(SPACE), num_query_points=q)
    model, data = _model()
    for rule in (AsynchronousOptimization(BatchMonteCarloExpectedImprovement(50)), AsynchronousGreedy(LocalPenalization(SPACE))):
        with pytest.raises(ValueError):
            rule.acquire(SPACE, {"foo": model}, {"foo": data})
        with pytest.raises(ValueError):
            rule.acquire(SPACE, {OBJECTIVE: model}, None)


def _rules():
    opt = generate_random_search_optimizer(300, seed=9, on_device=False)
    cont = generate_continuous_optimizer(num_initial_samples=200, num_optimization_runs=3)
    return [("qEI", lambda: AsynchronousOptimization(BatchMonteCarloExpectedImprovement(64), optimizer=opt)),
            ("LP", lambda: AsynchronousGreedy(LocalPenalization(SPACE, num_samples=100), optimizer=cont)),
            ("Fantasizer", lambda: AsynchronousGreedy(Fantasizer(), optimizer=cont)),
            ("GIBBON", lambda: AsynchronousGreedy(GIBBON(SPACE, grid_size=80), optimizer=cont))]


@pytest.mark.parametrize("name,make", _rules(), ids=[r[0] for r in _rules()])
def test_async_keeps_track_of_pending_points(name, make):
    model, data = _model()  # :528-571
    rule = make()
    state_fn = rule.acquire_single(SPACE, model, dataset=data)
    state, point1 = state_fn(None)
    if name == "qEI":  # the reparametrization sampler is reset per acquire and then wants a fixed batch size
        state_fn = rule.acquire_single(SPACE, model, dataset=data)
    state, point2 = state_fn(state)
    assert point1.shape == (1, 2) and point2.shape == (1, 2)
    assert state is not None and len(state.pending_points) == 2
    # pretend we saw the observation of the first point
    seen = data + Dataset(point1, np.array([[1.0]]))
    model.update(seen)
    state_fn = rule.acquire_single(SPACE, model, dataset=seen)
    state, point3 = state_fn(state)
    assert len(state.pending_points) == 2
    np.testing.assert_allclose(state.pending_points, np.concatenate([point2, point3], axis=0))
    if name != "qEI":  # greedy builders keep the pending points apart
        assert np.linalg.norm(point2 - point3) > 1e-6


def test_async_optimization_evaluates_the_batch_function_on_pending_plus_candidate():
    model, data = _model()
    seen = []

    class Spy(BatchMonteCarloExpectedImprovement):
        def prepare_acquisition_function(self, model, dataset=None):
            fn = super().prepare_acquisition_function(model, dataset)

            def spy(x):
                seen.append(tuple(np.asarray(x).shape))
                return fn(x)

            return spy

        def update_acquisition_function(self, function, model, dataset=None):
            return function

    rule = AsynchronousOptimization(Spy(32), optimizer=generate_random_search_optimizer(50, seed=1, on_device=False),
                                    num_query_points=2)
    state, pts = rule.acquire_single(SPACE, model, dataset=data)(AsynchronousRuleState(np.array([[0.5, 0.5]])))
    assert pts.shape == (2, 2) and len(state.pending_points) == 3
    assert seen and all(s == (50, 3, 2) for s in seen)  # [N, P + B, D]


def test_async_greedy_batch_and_loops_thread_the_state():
    model, data = _model()
    cont = generate_continuous_optimizer(num_initial_samples=200, num_optimization_runs=3)
    rule = AsynchronousGreedy(LocalPenalization(SPACE, num_samples=100), optimizer=cont, num_query_points=3)
    state, pts = rule.acquire_single(SPACE, model, dataset=data)(None)
    assert pts.shape == (3, 2) and len(state.pending_points) == 3
    # Ask-Tell: the optimizer keeps the state; told points leave the pending set on the next ask
    ask_tell = AskTellOptimizer(SPACE, data, model, AsynchronousGreedy(LocalPenalization(SPACE, num_samples=100), optimizer=cont),
                                fit_model=False)
    p1 = ask_tell.ask()
    p2 = ask_tell.ask()
    assert len(ask_tell.acquisition_state.pending_points) == 2
    ask_tell.tell(Dataset(p1, OBJ.scaled_branin(p1)))
    p3 = ask_tell.ask()
    np.testing.assert_allclose(ask_tell.acquisition_state.pending_points, np.concatenate([p2, p3]))
    # BayesianOptimizer: synchronous loop, so nothing stays pending from one step to the next but the last request
    model2, data2 = _model(seed=1)
    res = BayesianOptimizer(lambda x: Dataset(x, OBJ.scaled_branin(x)), SPACE).optimize(
        3, data2, model2, AsynchronousGreedy(Fantasizer(), optimizer=cont, num_query_points=2), fit_model=False)
    final = res.final_result.unwrap()
    assert len(final.dataset) == 12 + 6
    assert final.acquisition_state is not None and len(final.acquisition_state.pending_points) == 2
    assert res.history[0].acquisition_state is None and res.history[1].acquisition_state is not None
