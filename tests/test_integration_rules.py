"""CPU integration test after the reference's tests/integration/test_bayesian_optimization.py:103-330
(``test_bayesian_optimizer_with_gpr_finds_minima_of_simple_quadratic``): every acquisition rule built here must
solve the simple quadratic on [0, 1]^2 from 10 random points in at most 6 steps -- best point within 5 % of the
minimiser (1, 1), best value within 5 % of -4 -- through the unmodified BO loop with model fitting at every step.
The engine is replaced at its boundary by the oracle-backed stand-in (tests/fakes.py); the same rules run on the
real engine in tests/test_gpu_host.py."""
import numpy as np
import pytest

import trieste_amd
import trieste_amd.extras as A
import trieste_amd.models as M
from tests.fakes import FakeEngine
from trieste_amd import objectives as OBJ
from trieste_amd.bayesian_optimizer import BayesianOptimizer, stop_at_minimum
from trieste_amd.data import Dataset
from trieste_amd.space import Box


@pytest.fixture(autouse=True)
def fake_engine(monkeypatch):
    monkeypatch.setattr(M, "GPEngine", FakeEngine)


SPACE = Box([0.0, 0.0], [1.0, 1.0])


def _opt():
    return A.generate_continuous_optimizer(num_initial_samples=400, num_optimization_runs=4)


RULES = [
    ("EfficientGlobalOptimization", lambda: A.EfficientGlobalOptimization(optimizer=_opt())),
    ("AugmentedExpectedImprovement", lambda: A.EfficientGlobalOptimization(A.AugmentedExpectedImprovement(), optimizer=_opt())),
    ("MonteCarloExpectedImprovement", lambda: A.EfficientGlobalOptimization(
        A.MonteCarloExpectedImprovement(500), optimizer=A.generate_random_search_optimizer(600, on_device=False))),
    ("MinValueEntropySearch", lambda: A.EfficientGlobalOptimization(
        A.MinValueEntropySearch(SPACE, grid_size=200, min_value_sampler=A.GumbelSampler(True)), optimizer=_opt())),
    ("BatchMonteCarloExpectedImprovement", lambda: A.EfficientGlobalOptimization(
        A.BatchMonteCarloExpectedImprovement(200), num_query_points=3,
        optimizer=A.generate_random_search_optimizer(400, on_device=False))),
    ("AsynchronousOptimization", lambda: A.AsynchronousOptimization(
        A.BatchMonteCarloExpectedImprovement(200), num_query_points=2,
        optimizer=A.generate_random_search_optimizer(400, on_device=False))),
    ("LocalPenalization", lambda: A.EfficientGlobalOptimization(A.LocalPenalization(SPACE, num_samples=100),
                                                               num_query_points=3, optimizer=_opt())),
    ("LocalPenalization/AsynchronousGreedy", lambda: A.AsynchronousGreedy(A.LocalPenalization(SPACE, num_samples=100),
                                                                         num_query_points=3, optimizer=_opt())),
    ("GIBBON", lambda: A.EfficientGlobalOptimization(
        A.GIBBON(SPACE, grid_size=200, min_value_sampler=A.GumbelSampler(True)), num_query_points=2, optimizer=_opt())),
    ("MultipleOptimismNegativeLowerConfidenceBound", lambda: A.EfficientGlobalOptimization(
        A.MultipleOptimismNegativeLowerConfidenceBound(SPACE), num_query_points=3, optimizer=_opt())),
    ("TREGO", lambda: A.BatchTrustRegionBox(A.TREGOBox(SPACE), A.EfficientGlobalOptimization(optimizer=_opt()))),
    ("TREGO/MinValueEntropySearch", lambda: A.BatchTrustRegionBox(A.TREGOBox(SPACE), A.EfficientGlobalOptimization(
        A.MinValueEntropySearch(SPACE, grid_size=200, min_value_sampler=A.GumbelSampler(True)), optimizer=_opt()))),
    ("Turbo", lambda: A.BatchTrustRegionBox(A.TURBOBox(SPACE), A.DiscreteThompsonSampling(300, 1))),
    ("BatchTrustRegionBox", lambda: A.BatchTrustRegionBox([A.SingleObjectiveTrustRegionBox(SPACE) for _ in range(2)],
                                                         A.EfficientGlobalOptimization(optimizer=_opt()))),
    ("DiscreteThompsonSampling", lambda: A.DiscreteThompsonSampling(300, 3)),
    ("DiscreteThompsonSampling/trajectories", lambda: A.DiscreteThompsonSampling(
        300, 3, thompson_sampler=A.ThompsonSamplerFromTrajectory())),
    ("Fantasizer", lambda: A.EfficientGlobalOptimization(A.Fantasizer(), num_query_points=3, optimizer=_opt())),
    ("GreedyContinuousThompsonSampling", lambda: A.EfficientGlobalOptimization(
        A.GreedyContinuousThompsonSampling(), num_query_points=3, optimizer=_opt())),
    ("ParallelContinuousThompsonSampling", lambda: A.EfficientGlobalOptimization(
        A.ParallelContinuousThompsonSampling(), num_query_points=3, optimizer=_opt())),
]


@pytest.mark.parametrize("name,make_rule", RULES, ids=[r[0] for r in RULES])
def test_bayesian_optimizer_with_gpr_finds_minima_of_simple_quadratic(name, make_rule):
    trieste_amd.set_seed(1793)
    problem = OBJ.SimpleQuadratic
    initial = SPACE.sample(10, seed=7)
    data = Dataset(initial, problem.objective(initial))
    model = M.GaussianProcessRegression(M.build_gpr(data, SPACE, likelihood_variance=1e-7))
    num_steps = 6
    result = BayesianOptimizer(lambda x: Dataset(x, problem.objective(x)), SPACE).optimize(
        num_steps, data, model, make_rule(), fit_initial_model=False,
        early_stop_callback=stop_at_minimum(problem.minimum, problem.minimizers, minimum_rtol=0.05, minimum_step_number=2))
    assert result.final_result.is_ok, result.final_result
    assert 1 <= len(result.history) <= num_steps
    best_x, best_y, _ = result.try_get_optimal_point()
    minimizer_err = np.abs((best_x - problem.minimizers) / problem.minimizers)
    assert np.any(np.all(minimizer_err < 0.05, axis=-1)), (name, best_x)
    np.testing.assert_allclose(best_y, problem.minimum, rtol=0.05)
    assert len(result.try_get_final_dataset()) > 10


BRANIN_SPACE = OBJ.ScaledBranin.search_space
BRANIN_RULES = [  # (steps, id, rule): the reference's GPR_OPTIMIZER_PARAMS (test_bayesian_optimization.py:103-290)
    (20, "EfficientGlobalOptimization", lambda: A.EfficientGlobalOptimization()),
    (30, "AugmentedExpectedImprovement", lambda: A.EfficientGlobalOptimization(A.AugmentedExpectedImprovement())),
    (20, "MonteCarloExpectedImprovement", lambda: A.EfficientGlobalOptimization(
        A.MonteCarloExpectedImprovement(1000), A.generate_continuous_optimizer(100))),
    (24, "MinValueEntropySearch", lambda: A.EfficientGlobalOptimization(A.MinValueEntropySearch(
        BRANIN_SPACE, min_value_sampler=A.ThompsonSamplerFromTrajectory(sample_min_value=True)))),
    (12, "BatchMonteCarloExpectedImprovement", lambda: A.EfficientGlobalOptimization(
        A.BatchMonteCarloExpectedImprovement(500), num_query_points=3)),
    (15, "LocalPenalization", lambda: A.EfficientGlobalOptimization(A.LocalPenalization(BRANIN_SPACE), num_query_points=3)),
    (15, "LocalPenalization/AsynchronousGreedy", lambda: A.AsynchronousGreedy(A.LocalPenalization(BRANIN_SPACE),
                                                                              num_query_points=3)),
    (10, "GIBBON", lambda: A.EfficientGlobalOptimization(A.GIBBON(BRANIN_SPACE), num_query_points=2)),
    (25, "MultipleOptimismNegativeLowerConfidenceBound", lambda: A.EfficientGlobalOptimization(
        A.MultipleOptimismNegativeLowerConfidenceBound(BRANIN_SPACE), num_query_points=3)),
    (20, "TREGO", lambda: A.BatchTrustRegionBox(A.TREGOBox(BRANIN_SPACE))),
    (15, "TREGO/MinValueEntropySearch", lambda: A.BatchTrustRegionBox(
        A.TREGOBox(BRANIN_SPACE), A.EfficientGlobalOptimization(A.MinValueEntropySearch(BRANIN_SPACE)))),
    (20, "TREGO/ParallelContinuousThompsonSampling", lambda: A.BatchTrustRegionBox(
        [A.TREGOBox(BRANIN_SPACE) for _ in range(3)],
        A.EfficientGlobalOptimization(A.ParallelContinuousThompsonSampling(), num_query_points=3))),
    (10, "Turbo", lambda: A.BatchTrustRegionBox(A.TURBOBox(BRANIN_SPACE), A.DiscreteThompsonSampling(500, 3))),
    (10, "BatchTrustRegionBox", lambda: A.BatchTrustRegionBox(
        [A.SingleObjectiveTrustRegionBox(BRANIN_SPACE) for _ in range(3)],
        A.EfficientGlobalOptimization(A.ParallelContinuousThompsonSampling(), num_query_points=3))),
    (15, "DiscreteThompsonSampling", lambda: A.DiscreteThompsonSampling(500, 5)),
    (15, "Fantasizer", lambda: A.EfficientGlobalOptimization(A.Fantasizer(), num_query_points=3)),
    (10, "GreedyContinuousThompsonSampling", lambda: A.EfficientGlobalOptimization(
        A.GreedyContinuousThompsonSampling(), num_query_points=5)),
    (10, "ParallelContinuousThompsonSampling", lambda: A.EfficientGlobalOptimization(
        A.ParallelContinuousThompsonSampling(), num_query_points=5)),
]


@pytest.mark.slow  # as in the reference: run with --runslow yes (the step budgets are tuned to ITS seeds)
@pytest.mark.parametrize("num_steps,name,make_rule", BRANIN_RULES, ids=[r[1] for r in BRANIN_RULES])
def test_bayesian_optimizer_with_gpr_finds_minima_of_scaled_branin(num_steps, name, make_rule):
    """The reference's headline integration test (its ``slow`` tier, test_bayesian_optimization.py:300-316) with its own
    step budgets and defaults: from 5 random points every rule must bring the best observation within 0.5 % of the
    scaled Branin minimum and the best point within 5 % of one of the three minimisers."""
    trieste_amd.set_seed(1793)
    problem = OBJ.ScaledBranin
    initial = BRANIN_SPACE.sample(5, seed=1793)
    data = Dataset(initial, problem.objective(initial))
    model = M.GaussianProcessRegression(M.build_gpr(data, BRANIN_SPACE, likelihood_variance=1e-7))
    result = BayesianOptimizer(lambda x: Dataset(x, problem.objective(x)), BRANIN_SPACE).optimize(
        num_steps, data, model, make_rule(), fit_initial_model=False,
        early_stop_callback=stop_at_minimum(problem.minimum, problem.minimizers, minimum_rtol=0.005, minimum_step_number=2))
    assert result.final_result.is_ok, result.final_result
    best_x, best_y, _ = result.try_get_optimal_point()
    minimizer_err = np.abs((best_x - problem.minimizers) / problem.minimizers)
    assert np.any(np.all(minimizer_err < 0.05, axis=-1)), (name, best_x, best_y)
    np.testing.assert_allclose(best_y, problem.minimum, rtol=0.005)
