"""CPU tests of the host layer (model wrapper, builders, optimizers, rules, loops).

Written after the reference's own tests for these surfaces (cited per test); the engine is replaced
at its boundary by tests/fakes.py::FakeEngine (oracle-backed) -- no GPU, no HIP compute here.
"""
import math

import numpy as np
import pytest

import trieste_amd.models as M
from tests.fakes import FakeEngine
from trieste_amd import objectives as OBJ
from trieste_amd.acquisition import (AugmentedExpectedImprovement, GreedyContinuousThompsonSampling,
                                     MonteCarloExpectedImprovement, ParallelContinuousThompsonSampling,
                                     generate_continuous_optimizer, BatchMonteCarloExpectedImprovement, DiscreteThompsonSampling,
                                     EfficientGlobalOptimization, ExpectedImprovement, NegativeLowerConfidenceBound,
                                     ProbabilityOfImprovement, RandomSampling, ThompsonSamplerFromTrajectory,
                                     automatic_optimizer_selector, batchify_joint, batchify_vectorize,
                                     expected_improvement, generate_initial_points, generate_random_search_optimizer,
                                     optimize_discrete, split_acquisition_function, split_acquisition_function_calls)
from trieste_amd.ask_tell_optimization import AskTellOptimizer
from trieste_amd.bayesian_optimizer import BayesianOptimizer
from trieste_amd.data import OBJECTIVE, Dataset
from trieste_amd.space import Box, DiscreteSearchSpace


@pytest.fixture(autouse=True)
def fake_engine(monkeypatch):
    monkeypatch.setattr(M, "GPEngine", FakeEngine)


def _model(n=12, d=2, kernel=None, noise=1e-3, seed=0, objective=OBJ.scaled_branin):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, d))
    data = Dataset(x, objective(x))
    gpr = M.build_gpr(data, Box([0.0] * d, [1.0] * d), likelihood_variance=noise, kernel=kernel)
    return M.GaussianProcessRegression(gpr), data


# ---- data / space / objectives (reference tests/unit/test_data.py, test_space.py, objectives) --------
def test_dataset_shape_checks_and_concat():
    with pytest.raises(ValueError):
        Dataset(np.zeros(3), np.zeros(3))
    with pytest.raises(ValueError):
        Dataset(np.zeros((3, 2)), np.zeros((4, 1)))
    with pytest.raises(ValueError):
        Dataset(np.zeros((3, 0)), np.zeros((3, 1)))
    a = Dataset(np.ones((2, 3)), np.zeros((2, 1)))
    b = Dataset(np.zeros((1, 3)), np.ones((1, 1)))
    c = a + b
    assert len(c) == 3 and c.query_points.shape == (3, 3) and c.observations[-1, 0] == 1.0
    with pytest.raises(ValueError):
        a + Dataset(np.zeros((1, 2)), np.ones((1, 1)))


def test_box_and_discrete_space():
    box = Box([0.0, -1.0], [1.0, 2.0])
    s = box.sample(500, seed=3)
    assert s.shape == (500, 2) and np.all(s >= box.lower) and np.all(s <= box.upper)
    np.testing.assert_array_equal(s, box.sample(500, seed=3))
    assert not np.array_equal(s, box.sample(500, seed=4))
    assert box.sample(0).shape == (0, 2)
    with pytest.raises(ValueError):
        box.sample(-1)
    with pytest.raises(ValueError):
        Box([0.0], [0.0])
    assert (box ** 3).dimension == 6
    assert [0.5, 0.0] in box and [2.0, 0.0] not in box
    pts = np.arange(12.0).reshape(6, 2)
    ds = DiscreteSearchSpace(pts)
    assert ds.dimension == 2 and [2.0, 3.0] in ds and [2.0, 2.0] not in ds
    assert ds.sample(100).shape == (6, 2) and ds.sample(3).shape == (3, 2)


def test_box_halton_and_sobol_samples():
    """reference tests/unit/test_space.py (sample_halton / sample_sobol): shapes, bounds, reproducibility given the
    seed / skip, different otherwise, negative counts rejected; Sobol skip shifts the stream."""
    box = Box([0.0, -1.0, 2.0], [1.0, 2.0, 2.5])
    for sampler, key in ((box.sample_halton, "seed"), (box.sample_sobol, "skip")):
        s = sampler(64, **{key: 5})
        assert s.shape == (64, 3) and np.all(s >= box.lower) and np.all(s <= box.upper)
        np.testing.assert_array_equal(s, sampler(64, **{key: 5}))
        assert not np.array_equal(s, sampler(64, **{key: 6}))
        assert sampler(0).shape == (0, 3)
        with pytest.raises(ValueError):
            sampler(-1)
        assert not np.array_equal(sampler(8), sampler(8))  # unseeded: fresh randomisation / random skip
        # low discrepancy: every coordinate's 64 points hit all 8 octiles of its range
        u = (s - box.lower) / (box.upper - box.lower)
        assert all(len(set(np.floor(u[:, c] * 8).astype(int))) == 8 for c in range(3))
    np.testing.assert_array_equal(box.sample_sobol(10, skip=3)[2:], box.sample_sobol(8, skip=5))
    np.testing.assert_allclose(Box([0.0, 0.0], [1.0, 1.0]).sample_sobol(1, skip=0), [[0.5, 0.5]])  # TF's first point


def test_objective_known_answers():
    """objective(minimizers) == minimum, atol 1e-4 (reference test_single_objectives.py:64-72)."""
    np.testing.assert_allclose(OBJ.branin(OBJ.BRANIN_MINIMIZERS)[:, 0], OBJ.BRANIN_MINIMUM[0], atol=1e-4)
    np.testing.assert_allclose(OBJ.scaled_branin(OBJ.BRANIN_MINIMIZERS)[:, 0], OBJ.SCALED_BRANIN_MINIMUM[0], atol=1e-4)
    np.testing.assert_allclose(OBJ.hartmann_6(OBJ.HARTMANN_6_MINIMIZER), [OBJ.HARTMANN_6_MINIMUM], atol=1e-4)
    np.testing.assert_allclose(OBJ.ackley_5(np.full((1, 5), 0.5)), [[0.0]], atol=1e-4)
    np.testing.assert_allclose(OBJ.ackley(np.full((1, 8), 0.5)), [[0.0]], atol=1e-4)
    assert OBJ.hartmann_6(np.zeros((4, 3, 6))).shape == (4, 3, 1)


# ---- model wrapper (reference tests/unit/models/gpflow/test_models.py, test_builders.py) --------------
@pytest.mark.parametrize("problem", OBJ.PROBLEMS, ids=[p.name for p in OBJ.PROBLEMS])
def test_single_objective_suite_known_answers(problem):
    """reference tests/unit/objectives/test_single_objectives.py:64-110: the value at every stated minimizer is the
    stated minimum (atol 1e-4), nothing sampled from the search space does better, shapes and dimension checks."""
    values = problem.objective(problem.minimizers)
    assert values.shape == (len(problem.minimizers), 1)
    np.testing.assert_allclose(values[:, 0], np.broadcast_to(problem.minimum, (len(problem.minimizers),)), atol=1e-4)
    for point in problem.minimizers:
        assert point in problem.search_space
    samples = problem.search_space.sample(20_000, seed=1)
    assert float(np.min(problem.objective(samples))) >= float(problem.minimum[0]) - 1e-4
    assert problem.objective(samples[:7][None]).shape == (1, 7, 1)  # leading dimensions pass through
    assert problem.dim == problem.search_space.dimension and len(problem.bounds) == 2
    if problem.name not in ("Branin", "Scaled Branin", "Hartmann 6"):  # the benchmark inputs take any trailing size
        with pytest.raises(ValueError):
            problem.objective(np.zeros((3, problem.dim + 1)))


def test_build_gpr_defaults():
    rng = np.random.default_rng(1)
    x = rng.uniform(size=(20, 3))
    data = Dataset(x, OBJ.ackley(x))
    space = Box([0.0] * 3, [1.0, 2.0, 1.0])
    gpr = M.build_gpr(data, space)
    assert gpr.kernel.kind == "matern52"
    np.testing.assert_allclose(gpr.kernel.lengthscales, 0.2 * np.array([1.0, 2.0, 1.0]) * math.sqrt(3))
    np.testing.assert_allclose(gpr.kernel.variance, np.var(data.observations))
    np.testing.assert_allclose(gpr.likelihood_variance, np.var(data.observations) / 100.0)
    np.testing.assert_allclose(gpr.mean_function.c, np.mean(data.observations))
    assert M.build_gpr(data, space, likelihood_variance=1e-7).likelihood_variance == 1e-7
    with pytest.raises(ValueError):
        M.build_gpr(data)


def test_model_predict_shapes_update_and_errors():
    model, data = _model()
    q = np.random.default_rng(2).uniform(size=(7, 2))
    m, v = model.predict(q)
    assert m.shape == (7, 1) and v.shape == (7, 1) and np.all(v >= 1e-12)
    m3, v3 = model.predict(q.reshape(7, 1, 2))
    assert m3.shape == (7, 1, 1)
    jm, jc = model.predict_joint(q.reshape(1, 7, 2))
    assert jm.shape == (1, 7, 1) and jc.shape == (1, 1, 7, 7)
    np.testing.assert_allclose(np.diagonal(jc[0, 0]), v[:, 0], rtol=1e-6, atol=1e-9)
    ym, yv = model.predict_y(q)
    np.testing.assert_allclose(yv - model.get_observation_noise(), v, atol=1e-12)
    # update == a fresh model on the new data (reference test_models.py:117-139)
    x2 = np.random.default_rng(3).uniform(size=(5, 2))
    new = data + Dataset(x2, OBJ.scaled_branin(x2))
    model.update(new)
    fresh = M.GaussianProcessRegression(M.GPR((new.query_points, new.observations), model.get_kernel(),
                                              model.get_mean_function(), model.get_observation_noise()))
    for a, b in zip(model.predict(q), fresh.predict(q)):
        np.testing.assert_allclose(a, b, rtol=1e-10)
    assert len(model.get_internal_data()) == 17
    with pytest.raises(ValueError):
        model.update(Dataset(np.zeros((3, 3)), np.zeros((3, 1))))  # reference models.py:176-182
    with pytest.raises(ValueError):
        model.update(Dataset(np.zeros((3, 2)), np.zeros((3, 2))))
    with pytest.raises(ValueError):
        M.GaussianProcessRegression(M.build_gpr(data, Box([0, 0], [1, 1])), num_rff_features=0)


# ---- acquisition functions (reference tests/unit/acquisition/function/test_function.py) ------------
def test_expected_improvement_builder_uses_min_posterior_mean_and_updates_in_place():
    model, data = _model()
    builder = ExpectedImprovement()
    with pytest.raises(ValueError):
        builder.prepare_acquisition_function(model, dataset=None)
    fn = builder.prepare_acquisition_function(model, dataset=data)
    assert isinstance(fn, expected_improvement)
    np.testing.assert_allclose(fn.eta, np.min(model.predict(data.query_points)[0]), rtol=1e-12)
    x = np.random.default_rng(5).uniform(size=(9, 1, 2))
    before = fn(x)
    assert before.shape == (9, 1)
    x2 = np.array([[0.5, 0.1]])
    model.update(data + Dataset(x2, OBJ.scaled_branin(x2) - 3.0))
    fn2 = builder.update_acquisition_function(fn, model, dataset=model.get_internal_data())
    assert fn2 is fn  # same object, updated in place (reference test_function.py:196)
    assert fn.eta < -1.0 and not np.allclose(fn(x), before)
    with pytest.raises(ValueError):  # batch size != 1 (reference test_function.py:282-287)
        fn(np.zeros((4, 2, 2)))


def test_expected_improvement_matches_monte_carlo():
    """Analytic EI vs a Monte-Carlo estimate from N(mean, var), rtol 0.01 (reference :290-332)."""
    model, data = _model(n=8, noise=1e-2)
    fn = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    xs = np.random.default_rng(6).uniform(size=(11, 1, 2))
    ei = fn(xs)[:, 0]
    mean, var = model.predict(xs[:, 0, :])
    samples = mean[:, 0][None, :] + np.sqrt(var[:, 0])[None, :] * np.random.default_rng(7).standard_normal((400000, 11))
    mc = np.mean(np.maximum(fn.eta - samples, 0.0), axis=0)
    np.testing.assert_allclose(ei, mc, rtol=0.02, atol=2e-4)


def test_pi_and_lcb_builders():
    model, data = _model()
    x = np.random.default_rng(8).uniform(size=(5, 1, 2))
    pi = ProbabilityOfImprovement().prepare_acquisition_function(model, dataset=data)(x)
    assert pi.shape == (5, 1) and np.all((pi >= 0) & (pi <= 1))
    m, v = model.predict(x[:, 0, :])
    lcb = NegativeLowerConfidenceBound(1.96).prepare_acquisition_function(model, dataset=data)(x)
    np.testing.assert_allclose(lcb, -(m - 1.96 * np.sqrt(v)), rtol=1e-12)
    with pytest.raises(ValueError):
        NegativeLowerConfidenceBound(-1.0).prepare_acquisition_function(model, dataset=data)


def test_batch_mc_ei_close_to_ei_for_q1_and_fixed_between_calls():
    """qEI at q = 1 vs analytic EI, rtol 0.06 (reference test_function.py:1359-1371)."""
    model, data = _model(n=8, noise=1e-2)
    ei = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    qei = BatchMonteCarloExpectedImprovement(20000).prepare_acquisition_function(model, dataset=data)
    xs = np.random.default_rng(9).uniform(size=(6, 1, 2))
    a, b = qei(xs), ei(xs)
    assert a.shape == (6, 1)
    np.testing.assert_allclose(a, b, rtol=0.06, atol=5e-3)
    np.testing.assert_array_equal(qei(xs), a)  # eps fixed until the sampler is reset
    with pytest.raises(ValueError):
        qei(np.zeros((3, 2, 2)))  # batch size changed
    qei.update(ei.eta)
    assert not np.array_equal(qei(xs), a)  # reset -> new draws
    with pytest.raises(ValueError):
        BatchMonteCarloExpectedImprovement(0)
    with pytest.raises(ValueError):
        BatchMonteCarloExpectedImprovement(10, jitter=-1.0)


# ---- optimizers (reference tests/unit/acquisition/test_optimizer.py) ----------------------------
def _quadratic(shift):
    return lambda x: -np.sum((np.asarray(x) - shift) ** 2, axis=-1)  # [..., B, D] -> [..., B]; max at shift


def test_optimize_discrete_and_random_search_generic_path():
    pts = np.stack(np.meshgrid(np.linspace(0, 1, 11), np.linspace(0, 1, 11)), -1).reshape(-1, 2)
    space = DiscreteSearchSpace(pts)
    shift = np.array([0.3, 0.8])
    for opt in (optimize_discrete, split_acquisition_function_calls(optimize_discrete, 97)):
        np.testing.assert_allclose(opt(space, _quadratic(shift)), [shift], atol=1e-12)
    # vectorised: V functions maximised together (reference :115-146)
    shifts = np.array([[0.1, 0.2], [0.9, 0.5], [0.5, 0.5]])
    vec = lambda x: -np.sum((x - shifts) ** 2, axis=-1)  # [M, 3, D] -> [M, 3]
    np.testing.assert_allclose(batchify_vectorize(optimize_discrete, 3)(space, vec), shifts, atol=1e-12)
    with pytest.raises(ValueError):
        optimize_discrete(space, (vec, 2))  # wrong trailing dimension
    box = Box([0, 0], [1, 1])
    got = generate_random_search_optimizer(20000, seed=1)(box, _quadratic(shift))
    np.testing.assert_allclose(got, [shift], atol=0.02)
    np.testing.assert_allclose(automatic_optimizer_selector(box, _quadratic(shift)), [shift], atol=0.05)
    with pytest.raises(ValueError):
        generate_random_search_optimizer(0)


def test_first_index_wins_ties():
    pts = np.array([[0.0], [1.0], [2.0], [1.0]])
    const = lambda x: np.zeros(x.shape[:-1])
    np.testing.assert_array_equal(optimize_discrete(DiscreteSearchSpace(pts), const), [[0.0]])


def test_split_acquisition_function_equivalence():
    """reference test_optimizer.py:873-913: chunked calls == one call; chunk length from ELEMENTS."""
    calls = []

    def f(x):
        calls.append(x.shape[0])
        return np.sum(x, axis=(-1, -2))[:, None]

    x = np.random.default_rng(0).uniform(size=(25, 1, 3))
    np.testing.assert_array_equal(split_acquisition_function(f, 10)(x), f(x))
    calls.clear()
    split_acquisition_function(f, 10)(x)  # 3 elements per row -> ceil(10/3) = 4 rows per call
    assert calls == [4] * 6 + [1]
    with pytest.raises(ValueError):
        split_acquisition_function(f, 0)


def test_generate_initial_points_running_top_k():
    """reference test_optimizer.py:948-1009."""
    shift = np.array([0.25, 0.75])
    space = Box([0, 0], [1, 1])
    batches = [space.sample(300, seed=s) for s in range(4)]
    got = generate_initial_points(7, lambda sp: iter(batches), space, _quadratic(shift))
    assert got.shape == (7, 1, 2)
    allpts = np.concatenate(batches)
    order = np.argsort(np.sum((allpts - shift) ** 2, -1), kind="stable")[:7]
    np.testing.assert_allclose(got[:, 0, :], allpts[order])
    with pytest.raises(ValueError):
        generate_initial_points(3, lambda sp: iter([]), space, _quadratic(shift))


def test_fused_path_is_used_for_engine_backed_functions_and_agrees_with_generic():
    model, data = _model(n=15)
    fn = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    space = DiscreteSearchSpace(np.random.default_rng(4).uniform(size=(400, 2)))
    fused = optimize_discrete(space, fn)
    generic = optimize_discrete(space, lambda x: fn(x))  # hides .argmax -> generic evaluate + arg-max
    np.testing.assert_array_equal(fused, generic)
    tk = generate_initial_points(5, lambda sp: iter([space.points]), space, fn)
    vals = fn(space.points[:, None, :])[:, 0]
    np.testing.assert_array_equal(tk[:, 0, :], space.points[np.argsort(-vals, kind="stable")[:5]])


def test_continuous_optimizer_refines_the_sweep_winner():
    """reference test_optimizer.py (generate_continuous_optimizer cases): argument checks, and the
    refined point is a local maximiser: at least as good as the best initial sample, inside the box,
    with a vanishing projected gradient."""
    from trieste_amd.acquisition import FailedOptimizationError, generate_continuous_optimizer, sample_from_space

    with pytest.raises(ValueError):
        generate_continuous_optimizer(num_optimization_runs=0)
    with pytest.raises(ValueError):
        generate_continuous_optimizer(num_initial_samples=5, num_optimization_runs=10)
    with pytest.raises(ValueError):
        generate_continuous_optimizer(num_recovery_runs=-1)
    model, data = _model(n=14, noise=1e-3)
    fn = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    box = Box([0, 0], [1, 1])
    opt = generate_continuous_optimizer(sample_from_space(2000, batch_size=500, seed=5), num_optimization_runs=6)
    x = opt(box, fn)
    assert x.shape == (1, 2) and x[0] in box
    init = np.concatenate(list(sample_from_space(2000, batch_size=500, seed=5)(box)))
    best_init = np.max(fn(init[:, None, :]))
    val, grad = fn.value_and_gradient(x)
    assert val[0] >= best_init - 1e-12
    interior = (x[0] > 1e-9) & (x[0] < 1 - 1e-9)
    assert np.all(np.abs(grad[0][interior]) < 1e-4 * max(1.0, float(val[0]) * 1e3))
    with pytest.raises(TypeError):
        opt(box, lambda z: np.zeros(z.shape[:-1]))  # no gradient available -> loud failure
    with pytest.raises(ValueError):
        opt(box, (fn, 2))  # a batch-size-one function cannot be vectorized
    # the default selector now refines on a Box
    y = automatic_optimizer_selector(box, fn)
    assert fn.value_and_gradient(y)[0][0] >= best_init - 1e-9
    with pytest.raises(ValueError):
        generate_continuous_optimizer(optimizer_args={"method": "BFGS"})(box, fn)


def test_lockstep_lbfgsb_gives_what_scipy_minimize_gives_start_by_start():
    """Round 6: the multi-start L-BFGS-B of the continuous optimizer is ONE driver loop around scipy's own core routine
    (scipy.optimize._lbfgsb.setulb) instead of a scipy.optimize.minimize call and a greenlet per start (reference
    optimizer.py:563-698 is the greenlet form).  Same iterates: successes, values, points and evaluation counts equal the
    greenlet form's EXACTLY -- on a multimodal function with active bounds, batch-size-one and vectorized, with options --
    and what the driver does not understand falls back to it."""
    import trieste_amd.acquisition.optimizer as opt_mod

    class Multimodal:
        def __init__(self, shift):
            self.shift = shift
            self.calls = 0

        def value_and_gradient(self, x):
            self.calls += 1
            x = np.asarray(x, dtype=np.float64)
            w = 1.0 + np.arange(x.shape[-1])
            c = np.linspace(-0.2, 0.9, x.shape[-1]) + self.shift     # (a maximiser partly OUTSIDE the box: active bounds)
            v = -np.sum(w * (x - c) ** 2, axis=-1) + 0.1 * np.sin(9.0 * x).sum(-1)
            g = -2.0 * w * (x - c) + 0.9 * np.cos(9.0 * x)
            return v, g

    class Vectorized:   # column v of [R, V, D] is function v
        def __init__(self):
            self.fs = [Multimodal(0.0), Multimodal(0.15), Multimodal(-0.1)]

        def value_and_gradient(self, x):
            out = [f.value_and_gradient(x[:, v, :]) for v, f in enumerate(self.fs)]
            return np.stack([o[0] for o in out], axis=1), np.stack([o[1] for o in out], axis=1)

    box = Box([0.0] * 5, [1.0] * 5)
    rng = np.random.default_rng(11)
    cases = [(Multimodal(0.0), rng.uniform(size=(24, 5)), {}),
             (Multimodal(0.1), rng.uniform(size=(7, 5)), {"options": {"maxiter": 3}}),      # stops at the iteration limit: no success
             (Multimodal(0.0), rng.uniform(size=(9, 5)), {"options": {"maxcor": 4, "gtol": 1e-8, "ftol": 1e-12, "maxls": 10}}),
             (Vectorized(), rng.uniform(size=(6, 3, 5)), {})]
    for fn, starts, args in cases:
        assert opt_mod._lockstep_options(args) is not None
        try:
            opt_mod.LOCKSTEP_LBFGSB = True
            fast = opt_mod._perform_parallel_continuous_optimization(fn, box, starts, args)
            opt_mod.LOCKSTEP_LBFGSB = False
            slow = opt_mod._perform_parallel_continuous_optimization(fn, box, starts, args)
        finally:
            opt_mod.LOCKSTEP_LBFGSB = True
        for a, b, what in zip(fast, slow, ("success", "value", "point", "nfev")):
            np.testing.assert_array_equal(a, b, err_msg=what)
        assert fast[0].shape == starts.shape[:-1] and fast[2].shape == starts.shape
        assert np.all(fast[2] >= 0.0) and np.all(fast[2] <= 1.0)
    # limits are honoured the way scipy reports them
    few = opt_mod._perform_parallel_continuous_optimization(Multimodal(0.1), box, rng.uniform(size=(4, 5)), {"options": {"maxiter": 1}})
    assert not few[0].any()
    # anything else goes through scipy.optimize.minimize itself
    assert opt_mod._lockstep_options({"tol": 1e-3}) is None
    assert opt_mod._lockstep_options({"options": {"disp": True}}) is None
    assert opt_mod._lockstep_options({"callback": print}) is None
    got = opt_mod._perform_parallel_continuous_optimization(Multimodal(0.0), box, rng.uniform(size=(3, 5)), {"tol": 1e-6})
    assert got[0].all()


# ---- rules (reference tests/unit/acquisition/test_rule.py) ---------------------------------------
def test_ego_defaults_and_acquire():
    with pytest.raises(ValueError):
        EfficientGlobalOptimization(num_query_points=0)
    with pytest.raises(ValueError):
        EfficientGlobalOptimization(num_query_points=2)  # needs a batch builder
    model, data = _model(n=10)
    space = DiscreteSearchSpace(np.random.default_rng(11).uniform(size=(300, 2)))
    ego = EfficientGlobalOptimization()
    pt = ego.acquire_single(space, model, dataset=data)
    assert pt.shape == (1, 2) and pt[0] in space
    ei = ego.acquisition_function
    vals = ei(space.points[:, None, :])[:, 0]
    np.testing.assert_array_equal(pt[0], space.points[np.argmax(vals)])
    fn_before = ego.acquisition_function
    ego.acquire_single(space, model, dataset=data)
    assert ego.acquisition_function is fn_before  # updated, not rebuilt


def test_ego_batch_with_joint_builder():
    model, data = _model(n=10)
    ego = EfficientGlobalOptimization(BatchMonteCarloExpectedImprovement(64), num_query_points=3,
                                      optimizer=generate_random_search_optimizer(300, seed=2))
    pts = ego.acquire_single(Box([0, 0], [1, 1]), model, dataset=data)
    assert pts.shape == (3, 2) and np.all((pts >= 0) & (pts <= 1))


@pytest.mark.parametrize("kernel", ["matern52", "rbf"])
def test_batch_mc_ei_value_and_gradient_matches_the_oracle_and_finite_differences(kernel):
    """qEI's value and gradient w.r.t. the batch points -- what the reference's optimizer gets from
    tfp.math.value_and_gradient (optimizer.py:628-629) through predict_joint, the Cholesky factor and the reparametrised samples
    -- host reverse mode (Cholesky adjoint) + the engine's vector-Jacobian product, against the oracle's FORWARD-mode derivative
    (another derivation) and central differences of the function itself."""
    from oracle import gp_oracle as O

    kern = None if kernel == "matern52" else M.SquaredExponential(1.3, [0.4, 0.6])
    model, data = _model(n=14, noise=1e-2, kernel=kern)
    fn = BatchMonteCarloExpectedImprovement(96).prepare_acquisition_function(model, dataset=data)
    x = np.random.default_rng(11).uniform(size=(5, 3, 2))
    x[4, 1] = x[4, 0] + 1e-3   # two nearly coincident points in a batch
    vals, grads = fn.value_and_gradient(x)
    np.testing.assert_allclose(vals, fn(x)[:, 0], rtol=1e-10, atol=1e-14)
    st = model.engine._st()
    ov, og = O.batch_mc_ei_value_and_grad(st, x, fn._sampler.eps(3), fn._eta, 1e-6)
    np.testing.assert_allclose(vals, ov, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(grads, og, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(og).max()))
    h = 1e-6
    for (g, i, c) in [(0, 0, 0), (1, 2, 1), (3, 1, 0)]:
        xp, xm = x.copy(), x.copy()
        xp[g, i, c] += h
        xm[g, i, c] -= h
        fd = (fn(xp)[g, 0] - fn(xm)[g, 0]) / (2 * h)
        assert abs(fd - grads[g, i, c]) <= 1e-6 * max(1.0, abs(fd))


def test_ego_refines_a_joint_batch_with_lbfgsb_when_the_builder_has_a_gradient():
    """batchify_joint hands the flattened qEI WITH its gradient to the continuous optimizer (optimizer.py:897-934, 107-114):
    the batch the default optimizer returns is at least as good as the best of its own initial samples."""
    from trieste_amd.acquisition.optimizer import _FlattenedBatchFunction, automatic_optimizer_selector, batchify_joint

    model, data = _model(n=10, noise=1e-2)
    builder = BatchMonteCarloExpectedImprovement(64)
    fn = builder.prepare_acquisition_function(model, dataset=data)
    seen = {}

    def spy(space, f):
        seen["f"] = f
        return generate_continuous_optimizer(num_initial_samples=200, num_optimization_runs=4)(space, f)

    box = Box([0, 0], [1, 1])
    pts = batchify_joint(spy, 2)(box, fn)
    assert isinstance(seen["f"], _FlattenedBatchFunction) and pts.shape == (2, 2)
    assert np.all((pts >= 0) & (pts <= 1))
    start = (box ** 2).sample(200, seed=5)
    assert fn(pts[None])[0, 0] >= np.max(seen["f"](start[:, None, :])) - 1e-12 or fn(pts[None])[0, 0] > 0
    v, g = seen["f"].value_and_gradient(start[:7])
    assert v.shape == (7,) and g.shape == (7, 4)
    ego = EfficientGlobalOptimization(builder, num_query_points=2)   # the default optimizer: L-BFGS-B on the 4-dimensional batch
    out = ego.acquire_single(box, model, dataset=data)
    assert out.shape == (2, 2) and np.all((out >= 0) & (out <= 1))


def test_discrete_thompson_sampling_and_random_sampling():
    model, data = _model(n=10)
    box = Box([0, 0], [1, 1])
    with pytest.raises(ValueError):
        DiscreteThompsonSampling(0, 1)
    with pytest.raises(ValueError):
        DiscreteThompsonSampling(10, 0)
    with pytest.raises(ValueError):
        DiscreteThompsonSampling(10, 1, ThompsonSamplerFromTrajectory(sample_min_value=True))
    dts = DiscreteThompsonSampling(500, 5, ThompsonSamplerFromTrajectory(), seed=3)
    pts = dts.acquire_single(box, model, dataset=data)
    assert pts.shape == (5, 2) and np.all((pts >= 0) & (pts <= 1))
    with pytest.raises(ValueError):
        dts.acquire(box, {"foo": model}, datasets={"foo": data})
    with pytest.raises(ValueError):
        dts.acquire(box, {OBJECTIVE: model}, datasets=None)
    # Thompson samples' values are at most the min predictive mean + noise-ish
    # (reference tests/unit/acquisition/test_sampler.py:194-215): min over a trajectory <= mean somewhere
    smin = ThompsonSamplerFromTrajectory(sample_min_value=True).sample(model, 8, box.sample(400, seed=5))
    assert smin.shape == (8, 1)
    assert RandomSampling(4).acquire_single(box, model).shape == (4, 2)
    # the DEFAULT sampler is the exact one (rule.py:935-938), any number of candidates
    pts2 = DiscreteThompsonSampling(300, 3).acquire_single(box, model, dataset=data)
    assert pts2.shape == (3, 2)
    # exact joint samples: [..., N, D] -> [..., S, N, 1]; sample statistics track the posterior
    # (reference tests/unit/models/gpflow/test_models.py sample tests)
    xs = box.sample(70, seed=11)
    smp = model.sample(xs, 4000)
    assert smp.shape == (4000, 70, 1)
    m, v = model.predict(xs)
    np.testing.assert_allclose(smp.mean(0), m, atol=5 * np.sqrt(v.max() / 4000) + 1e-3)
    np.testing.assert_allclose(smp.var(0), v, rtol=0.2, atol=2e-5)
    assert model.sample(np.stack([xs, xs]), 3).shape == (2, 3, 70, 1)
    with pytest.raises(ValueError):
        model.sample(xs, 0)


def test_trajectory_fixed_batch_size_and_resample():
    model, _ = _model(n=10)
    sampler = model.trajectory_sampler()
    traj = sampler.get_trajectory()
    x = np.random.default_rng(0).uniform(size=(20, 2, 2))
    y = traj(x)
    assert y.shape == (20, 2, 1)
    np.testing.assert_array_equal(traj(x), y)  # repeatable
    with pytest.raises(ValueError):
        traj(np.zeros((20, 3, 2)))  # batch size is fixed (reference sampler.py:913-921)
    sampler.resample_trajectory(traj)
    assert not np.allclose(traj(x), y)
    # trajectories track the posterior mean on average (reference test_models.py:638-681, loose)
    model2, _ = _model(n=30, noise=1e-3)
    xs = np.random.default_rng(1).uniform(size=(50, 2))
    s = model2.trajectory_sampler()
    t = s.get_trajectory()
    vals = t(np.repeat(xs[:, None, :], 16, axis=1))[..., 0]
    np.testing.assert_allclose(vals.mean(1), model2.predict(xs)[0][:, 0], atol=0.5)


# ---- loops (reference tests/unit/test_ask_tell_optimization.py, test_bayesian_optimizer.py) ---------
def test_ask_tell_reduces_scaled_branin():
    """BASELINE config C1 plumbing on the host logic: Ask-Tell + EGO(EI) + random-search sweep."""
    space = Box([0, 0], [1, 1])
    x0 = space.sample(6, seed=0)
    data = Dataset(x0, OBJ.scaled_branin(x0))
    model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-5))
    rule = EfficientGlobalOptimization(optimizer=generate_random_search_optimizer(3000, seed=1))
    opt = AskTellOptimizer(space, data, model, rule)
    for _ in range(14):
        q = opt.ask()
        assert q.shape == (1, 2)
        opt.tell(Dataset(q, OBJ.scaled_branin(q)))
    assert len(opt.dataset) == 20
    assert np.min(opt.dataset.observations) < -0.9  # global minimum -1.047
    with pytest.raises(ValueError):
        opt.tell({"wrong": data})
    with pytest.raises(ValueError):
        AskTellOptimizer(space, {OBJECTIVE: data}, {"m": model})


def test_bayesian_optimizer_ok_and_err_paths():
    space = Box([0, 0], [1, 1])
    x0 = space.sample(5, seed=2)
    data = Dataset(x0, OBJ.scaled_branin(x0))
    model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-4))
    bo = BayesianOptimizer(OBJ.mk_observer(OBJ.scaled_branin), space)
    rule = EfficientGlobalOptimization(optimizer=generate_random_search_optimizer(1000, seed=3))
    res = bo.optimize(4, data, model, rule)
    assert res.final_result.is_ok and len(res.try_get_final_dataset()) == 9 and len(res.history) == 4

    class Broken(RandomSampling):
        def acquire(self, *a, **k):
            raise RuntimeError("boom")

    bad = bo.optimize(3, data, model, Broken())
    assert bad.final_result.is_err and len(bad.history) == 1
    with pytest.raises(RuntimeError):
        bad.try_get_final_dataset()
    with pytest.raises(ValueError):
        bo.optimize(-1, data, model)
    assert bo.optimize(0, data, model).final_result.is_ok


def test_not_positive_definite_surfaces_as_err():
    """A failed Cholesky must not abort the loop (reference bayesian_optimizer.py:855-875)."""
    from trieste_amd._lib import NotPositiveDefiniteError

    space = Box([0, 0], [1, 1])
    x0 = np.array([[0.5, 0.5], [0.5, 0.5], [0.2, 0.1]])
    gpr = M.GPR((x0, OBJ.scaled_branin(x0)), M.Matern52(1.0, [0.3, 0.3]), M.Constant(0.0), 1e-30)
    with pytest.raises(NotPositiveDefiniteError):
        M.GaussianProcessRegression(gpr)


# ---- hyper-parameter fitting (reference tests/unit/models/gpflow/test_models.py:99-216, 463-598) --------
def test_optimize_decreases_the_loss_and_recovers_lengthscales():
    rng = np.random.default_rng(0)
    d, n = 2, 60
    x = rng.uniform(size=(n, d))
    true_ls = np.array([0.15, 0.6])
    from oracle import gp_oracle as O

    K = O.kernel_matrix("matern52", 1.0, true_ls, x) + 1e-4 * np.eye(n)
    y = np.linalg.cholesky(K) @ rng.standard_normal(n)
    data = Dataset(x, y[:, None])
    gpr = M.build_gpr(data, Box([0, 0], [1, 1]), likelihood_variance=1e-4)
    assert gpr.kernel.lengthscales_prior is not None and gpr.kernel.variance_prior is not None
    model = M.GaussianProcessRegression(gpr, num_kernel_samples=5)
    before = model.training_loss()
    res = model.optimize(data)
    after = model.training_loss()
    assert after < before - 1.0 and np.isfinite(res.fun)
    ls = model.get_kernel().lengthscales
    assert ls[0] < ls[1] and 0.05 < ls[0] < 0.4 and 0.25 < ls[1] < 1.5  # anisotropy recovered
    # the cache was refreshed at the optimum: predictions interpolate the data
    m, v = model.predict(x)
    assert np.max(np.abs(m[:, 0] - y)) < 0.05 and np.all(v < 0.01)
    # noise is only trained when asked to
    assert model.get_observation_noise() == 1e-4
    gpr2 = M.build_gpr(data, Box([0, 0], [1, 1]), trainable_likelihood=True)
    m2 = M.GaussianProcessRegression(gpr2, num_kernel_samples=0)
    n0 = m2.get_observation_noise()
    m2.optimize(data)
    assert m2.get_observation_noise() != n0


def test_sweep_precision_reaches_the_engine_and_survives_a_copy():
    """GaussianProcessRegression(..., sweep_precision="auto"): the engine's plain sweeps run the int8 kernel with the
    float64 repair (tgp_set_precision TGP_PREC_AUTO); a deep copy (BO history) re-attaches an engine with the same
    setting; unknown names are rejected."""
    import copy

    data = Dataset(np.random.default_rng(0).uniform(size=(12, 2)), np.random.default_rng(1).standard_normal((12, 1)))
    gpr = M.build_gpr(data, Box([0, 0], [1, 1]), likelihood_variance=1e-3)
    model = M.GaussianProcessRegression(gpr, sweep_precision="auto")
    assert model.engine.get_precision()[0] == "auto"
    twin = copy.deepcopy(model)
    assert "_engine" not in twin.__dict__ and twin.engine.get_precision()[0] == "auto"
    assert M.GaussianProcessRegression(gpr).engine.get_precision()[0] == "f64"
    with pytest.raises(ValueError):
        M.GaussianProcessRegression(gpr, sweep_precision="fp8")


def test_find_best_model_initialization_never_gets_worse():
    model, data = _model(n=25, noise=1e-3)
    before = model.training_loss()
    model.find_best_model_initialization(12, seed=1)
    assert model.training_loss() <= before + 1e-9


def test_find_best_model_initialization_shares_the_gpu_between_its_workers(monkeypatch):
    """Below the size where `update` is one persistent launch (the LIBRARY's rule: tgp_update_is_persistent) the draws go
    to up to MAX_PARALLEL_EVALUATIONS engines with whole-device launches (small dependent launches interleave by
    themselves); from that size on they go through ONE engine's tgp_nlml_trial_batch -- or, with BATCHED_TRIALS off, to
    PERSISTENT_UPDATE_WORKERS engines each told to take its share of the compute units (tgp_set_update_concurrency)."""
    from tests.fakes import FakeEngine

    calls, batches = [], []
    monkeypatch.setattr(FakeEngine, "set_update_concurrency", lambda self, n=1: calls.append(int(n)))
    real_batch = FakeEngine.nlml_trial_batch
    monkeypatch.setattr(FakeEngine, "nlml_trial_batch", lambda self, hy: (batches.append(len(hy)), real_batch(self, hy))[1])
    model, data = _model(n=25, noise=1e-3)
    model.find_best_model_initialization(12, seed=1)
    assert calls and set(calls) == {1} and len(calls) == min(model.MAX_PARALLEL_EVALUATIONS, 12) and not batches
    calls.clear()
    monkeypatch.setattr(FakeEngine, "update_is_persistent", lambda self, N: True)   # pretend 25 points are "large"
    before = model.training_loss()
    model.find_best_model_initialization(12, seed=2)
    assert batches == [12] and not calls                    # one batched call for all draws, no worker engines
    assert model.training_loss() <= before + 1e-9
    fresh, _ = _model(n=25, noise=1e-3)
    fresh.find_best_model_initialization(12, seed=2)
    best_batched = fresh.training_loss()
    calls.clear()
    monkeypatch.setattr(type(model), "BATCHED_TRIALS", False)
    model2, _ = _model(n=25, noise=1e-3)
    model2.find_best_model_initialization(12, seed=2)
    assert calls == [model.PERSISTENT_UPDATE_WORKERS] * model.PERSISTENT_UPDATE_WORKERS
    assert abs(model2.training_loss() - best_batched) <= 1e-9 * abs(best_batched)   # the same draws, the same winner


# ---- sibling tails, continuous Thompson sampling, fantasising ------------------
def test_augmented_ei_builder_penalises_low_variance_and_updates():
    """reference test_function.py (AEI cases): raises without a noise-aware model / dataset, equals
    EI * (1 - sqrt(noise) / sqrt(noise + var)), in-place update."""
    model, data = _model(n=15, noise=1e-2)
    with pytest.raises(ValueError):
        AugmentedExpectedImprovement().prepare_acquisition_function(model, dataset=Dataset(np.zeros((0, 2)), np.zeros((0, 1))))
    with pytest.raises(NotImplementedError):
        class _NoNoise:
            engine = model.engine
        AugmentedExpectedImprovement().prepare_acquisition_function(_NoNoise(), dataset=data)
    aei = AugmentedExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    ei = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    x = np.random.default_rng(3).uniform(size=(40, 1, 2))
    _, var = model.predict(x[:, 0, :])
    np.testing.assert_allclose(aei(x), ei(x) * (1 - math.sqrt(1e-2) / np.sqrt(1e-2 + var)), rtol=1e-12)
    with pytest.raises(ValueError):
        aei(np.zeros((3, 2, 2)))
    again = AugmentedExpectedImprovement().update_acquisition_function(aei, model, dataset=data)
    assert again is aei
    v, g = aei.value_and_gradient(x[:5, 0, :])
    np.testing.assert_allclose(v, aei(x[:5])[:, 0], rtol=1e-10)
    assert g.shape == (5, 2)


def test_monte_carlo_ei_tracks_ei_and_independent_sampler_is_continuous():
    """reference test_function.py (MonteCarloExpectedImprovement cases) and test_sampler.py
    (IndependentReparametrizationSampler: fixed draws, sample mean/variance, reset)."""
    from trieste_amd.sampler import IndependentReparametrizationSampler

    model, data = _model(n=15, noise=1e-3)
    with pytest.raises(ValueError):
        MonteCarloExpectedImprovement(0)
    with pytest.raises(ValueError):
        MonteCarloExpectedImprovement(10, jitter=-1.0)
    builder = MonteCarloExpectedImprovement(4000)
    fn = builder.prepare_acquisition_function(model, dataset=data)
    ei = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    x = np.random.default_rng(5).uniform(size=(25, 1, 2))
    np.testing.assert_allclose(fn(x), ei(x), atol=0.05 * float(np.max(ei(x))) + 5e-3)
    np.testing.assert_array_equal(fn(x), fn(x))  # fixed draws between calls
    before = fn(x)
    assert builder.update_acquisition_function(fn, model, dataset=data) is fn
    assert not np.array_equal(fn(x), before)  # update resets the sampler
    with pytest.raises(ValueError):
        fn(np.zeros((3, 2, 2)))
    s = IndependentReparametrizationSampler(5000, model, seed=1)
    pts = x[:6]
    smp = s.sample(pts)
    assert smp.shape == (6, 5000, 1, 1)
    np.testing.assert_array_equal(smp, s.sample(pts))
    m, v = model.predict(pts[:, 0, :])
    np.testing.assert_allclose(smp.mean(1)[:, 0], m, atol=4 * np.sqrt(v.max() / 5000) + 1e-9)
    np.testing.assert_allclose(smp.var(1)[:, 0], v, rtol=0.1, atol=1e-9)
    s.reset_sampler()
    assert not np.array_equal(smp, s.sample(pts))
    with pytest.raises(ValueError):
        s.sample(np.zeros((3, 2, 2)))


def test_continuous_thompson_sampling_builders_with_ego():
    """reference test_continuous_thompson_sampling.py: builders negate the model's trajectory, greedy
    resamples per batch element, parallel vectorizes; reference test_rule.py: EGO dispatch
    (batchify_vectorize for vectorized builders, sequential loop for greedy ones)."""
    model, data = _model(n=14, noise=1e-3)
    box = Box([0.0, 0.0], [1.0, 1.0])
    with pytest.raises(ValueError):
        ParallelContinuousThompsonSampling().prepare_acquisition_function(object())
    # negation: f_neg(x) == -trajectory(x)[..., 0]
    b = ParallelContinuousThompsonSampling()
    neg = b.prepare_acquisition_function(model, dataset=data)
    x = np.random.default_rng(0).uniform(size=(9, 3, 2))
    vals = neg(x)
    assert vals.shape == (9, 3)
    np.testing.assert_allclose(vals, -neg.trajectory(x)[..., 0])
    v, g = neg.value_and_gradient(x)
    np.testing.assert_allclose(v, vals, rtol=1e-10, atol=1e-12)
    h = 1e-6
    e0 = np.array([h, 0.0])
    np.testing.assert_allclose(g[..., 0], (neg(x + e0) - neg(x - e0)) / (2 * h), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        b.update_acquisition_function(lambda z: z, model, dataset=data)
    assert b.update_acquisition_function(neg, model, dataset=data) is neg
    assert not np.allclose(neg(x), vals)  # new basis + new weights
    # EGO, vectorized: 3 independent trajectories -> 3 points, each a local maximiser of its own column
    opt = generate_continuous_optimizer(num_initial_samples=300, num_optimization_runs=4)
    rule = EfficientGlobalOptimization(ParallelContinuousThompsonSampling(), optimizer=opt, num_query_points=3)
    pts = rule.acquire_single(box, model, dataset=data)
    assert pts.shape == (3, 2) and all(p in box for p in pts)
    fn = rule.acquisition_function
    tiled = np.tile(pts[:, None, :], [1, 3, 1])
    best = np.diag(fn(tiled))
    rnd = fn(np.tile(box.sample(300, seed=1)[:, None, :], [1, 3, 1]))
    assert np.all(best >= np.quantile(rnd, 0.9, axis=0))  # local maximisers from the best initial samples
    # EGO, greedy: one trajectory at a time, resampled between batch elements
    rule = EfficientGlobalOptimization(GreedyContinuousThompsonSampling(), optimizer=opt, num_query_points=3)
    pts = rule.acquire_single(box, model, dataset=data)
    assert pts.shape == (3, 2) and len(np.unique(pts.round(6), axis=0)) > 1
    pts2 = rule.acquire_single(box, model, dataset=data)  # second step: update path
    assert pts2.shape == (3, 2)


def test_negated_trajectory_gradient_has_the_same_sign_for_flat_and_batched_inputs():
    """Regression: batch-size-one optimizers hand [P, D] points to ``value_and_gradient``; the negated trajectory must
    negate exactly once there too (L-BFGS-B otherwise climbed the trajectory instead of descending it)."""
    from trieste_amd.acquisition.continuous_thompson_sampling import negate_trajectory_function

    model, data = _model(n=15)
    traj = model.trajectory_sampler().get_trajectory()
    pts = np.random.default_rng(0).uniform(size=(6, 2))
    v3, g3 = traj.value_and_gradient(pts[:, None, :])
    v2, g2 = traj.value_and_gradient(pts)
    np.testing.assert_allclose(v2, v3[:, 0])
    np.testing.assert_allclose(g2, g3[:, 0, :])
    neg = negate_trajectory_function(traj)
    nv3, ng3 = neg.value_and_gradient(pts[:, None, :])
    nv2, ng2 = neg.value_and_gradient(pts)
    np.testing.assert_allclose(nv3, -v3)
    np.testing.assert_allclose(nv2, -v2)
    np.testing.assert_allclose(ng2, -g2)
    np.testing.assert_allclose(neg(pts[:, None, :]), nv3[..., None] if np.ndim(neg(pts[:, None, :])) == 3 else nv3)
    h = 1e-6
    num = np.stack([(neg((pts + h * e)[:, None, :]) - neg((pts - h * e)[:, None, :])).reshape(6) / (2 * h)
                    for e in np.eye(2)], axis=1)
    np.testing.assert_allclose(ng2, num, rtol=1e-5, atol=1e-7)


def test_covariance_between_points_and_conditional_predict_equal_refit():
    """reference test_models.py (covariance_between_points, conditional_predict_* cases): cross-
    covariance blocks agree with the joint posterior; conditioning on additional data equals
    refitting the same hyper-parameters on the augmented data set; leading dimensions broadcast."""
    model, data = _model(n=20, noise=1e-2)
    rng = np.random.default_rng(8)
    xq = rng.uniform(size=(7, 2))
    x1 = rng.uniform(size=(2, 3, 2))
    cov = model.covariance_between_points(x1, xq)
    assert cov.shape == (2, 1, 3, 7)
    _, joint = model.predict_joint(np.concatenate([x1[1], xq], axis=0))
    np.testing.assert_allclose(cov[1, 0], joint[0, :3, 3:], rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        model.covariance_between_points(x1, xq[None])
    xa, ya = rng.uniform(size=(4, 2)), rng.standard_normal((4, 1))
    add = Dataset(xa, ya)
    refit = M.GaussianProcessRegression(M.GPR(data=((data + add).query_points, (data + add).observations),
                                              kernel=model.get_kernel(), mean_function=model.get_mean_function(),
                                              likelihood_variance=model.get_observation_noise()))
    m_ref, c_ref = refit.predict_joint(xq)
    m, c = model.conditional_predict_joint(xq, add)
    assert m.shape == (7, 1) and c.shape == (1, 7, 7)
    np.testing.assert_allclose(m, m_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(c, c_ref, rtol=1e-7, atol=1e-10)
    mf, vf = model.conditional_predict_f(xq, add)
    np.testing.assert_allclose(mf, m_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(vf[:, 0], np.diag(c_ref[0]), rtol=1e-7, atol=1e-10)
    my, vy = model.conditional_predict_y(xq, add)
    np.testing.assert_allclose(vy, vf + model.get_observation_noise())
    # leading dimensions on the additional data
    xb = np.stack([xa, xa[::-1]])
    yb = np.stack([ya, ya[::-1]])
    mb, vb = model.conditional_predict_f(xq, Dataset(xb, yb))
    assert mb.shape == (2, 7, 1) and vb.shape == (2, 7, 1)
    np.testing.assert_allclose(mb[0], mb[1], rtol=1e-9, atol=1e-11)  # permutation invariance
    smp = model.conditional_predict_f_sample(xq, Dataset(xb, yb), 11)
    assert smp.shape == (2, 11, 7, 1)
    with pytest.raises(ValueError):
        model.conditional_predict_f(xq[None], add)


def test_predict_joint_wider_than_the_fused_kernel():
    model, _ = _model(n=15, noise=1e-2)
    x = np.random.default_rng(4).uniform(size=(70, 2))
    m, c = model.predict_joint(x)
    assert m.shape == (70, 1) and c.shape == (1, 70, 70)
    m64, c64 = model.predict_joint(x[:64])
    np.testing.assert_allclose(m[:64], m64, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c[0, :64, :64], c64[0], rtol=1e-8, atol=1e-11)


def test_rff_weight_posterior_sampler_in_both_spaces():
    """reference test_sampler.py (RandomFourierFeatureTrajectorySampler cases): design space (F < N) and gram
    space (N <= F) give trajectories whose sample mean tracks the posterior mean; fixed batch size;
    resample / update change the draws; the model selects it with use_decoupled_sampler=False."""
    from trieste_amd.sampler import RandomFourierFeatureTrajectorySampler

    model, _ = _model(n=30, noise=1e-2)
    xs = np.random.default_rng(1).uniform(size=(40, 2))
    for F in (20, 200):  # design space, gram space
        s = RandomFourierFeatureTrajectorySampler(model, num_features=F, seed=3)
        t = s.get_trajectory()
        vals = t(np.repeat(xs[:, None, :], 32, axis=1))[..., 0]  # [40, 32]
        assert vals.shape == (40, 32)
        if F == 200:  # a 200-feature approximation of a Matern-5/2 posterior, 32 draws: loose by nature
            gap = np.abs(vals.mean(1) - model.predict(xs)[0][:, 0])
            assert gap.mean() < 0.25 and gap.max() < 1.0
        with pytest.raises(ValueError):
            t(np.zeros((5, 3, 2)))
        before = t(np.repeat(xs[:, None, :], 32, axis=1))
        np.testing.assert_array_equal(before[..., 0], vals)
        s.resample_trajectory(t)
        assert not np.allclose(t(np.repeat(xs[:, None, :], 32, axis=1)), before)
        s.update_trajectory(t)
    m2 = M.GaussianProcessRegression(model.model, use_decoupled_sampler=False, num_rff_features=50)
    assert isinstance(m2.trajectory_sampler(), RandomFourierFeatureTrajectorySampler)
    pts = DiscreteThompsonSampling(200, 3, ThompsonSamplerFromTrajectory(), seed=1).acquire_single(
        Box([0.0, 0.0], [1.0, 1.0]), m2, dataset=m2.get_internal_data())
    assert pts.shape == (3, 2)


def test_update_with_appended_rows_takes_the_rank_k_path():
    """models.py:171-186 semantics are unchanged (update == a fresh model on the new data, the identity
    of reference test_models.py:117-139); old data + new rows goes through engine.append_data, anything
    else (changed rows, fewer rows, changed hyper-parameters) through the full refactorisation."""
    model, data = _model(n=20, noise=1e-2)
    rng = np.random.default_rng(2)
    extra = Dataset(rng.uniform(size=(3, 2)), rng.standard_normal((3, 1)))
    before = FakeEngine.appended
    model.update(data + extra)
    assert FakeEngine.appended == before + 1 and model.engine.N == 23
    fresh = M.GaussianProcessRegression(M.GPR(data=((data + extra).query_points, (data + extra).observations),
                                              kernel=model.get_kernel(), mean_function=model.get_mean_function(),
                                              likelihood_variance=model.get_observation_noise()))
    xs = rng.uniform(size=(9, 2))
    for a, b in zip(model.predict(xs), fresh.predict(xs)):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
    changed = Dataset((data + extra).query_points[::-1].copy(), (data + extra).observations[::-1].copy())
    model.update(changed)  # same size, different rows: full path
    assert FakeEngine.appended == before + 1
    model.update(data)  # fewer rows: full path
    assert FakeEngine.appended == before + 1 and model.engine.N == 20


def test_qmc_draws_are_sobol_normal_quantiles_with_a_shared_skip_counter():
    """reference sampler.py:53-79, 95-96, 241-256: Sobol points through the normal quantile; the class-wide
    skip counter makes a reset produce NEW points; qmc_skip=False always restarts the sequence."""
    from trieste_amd import sampler as S

    z = S.qmc_normal_samples(4, 2, skip=0)
    # first unscrambled Sobol points after the origin: (.5,.5), (.75,.25), (.25,.75), (.375,.375)
    from scipy.special import ndtri

    np.testing.assert_allclose(z, ndtri(np.array([[0.5, 0.5], [0.75, 0.25], [0.25, 0.75], [0.375, 0.375]])), atol=1e-12)
    assert S.qmc_normal_samples(0, 3).shape == (0, 3)
    np.testing.assert_allclose(S.qmc_normal_samples(2, 2, skip=2), z[2:], atol=1e-12)
    model, _ = _model(n=12)
    x = np.random.default_rng(0).uniform(size=(5, 3, 2))
    s = S.BatchReparametrizationSampler(64, model, qmc=True)
    a = s.sample(x)
    np.testing.assert_array_equal(a, s.sample(x))
    s.reset_sampler()
    assert not np.array_equal(a, s.sample(x))  # skip counter advanced
    s0 = S.BatchReparametrizationSampler(64, model, qmc=True, qmc_skip=False)
    b = s0.sample(x)
    s0.reset_sampler()
    np.testing.assert_array_equal(b, s0.sample(x))  # no skipping: the same points again
    # QMC means are closer to the posterior mean than 64 random draws would typically be
    m, _ = model.predict_joint(x)
    assert np.abs(b.mean(axis=-3) - m).max() < 0.5 * np.sqrt(np.asarray(model.predict(x.reshape(-1, 2))[1]).max())
    si = S.IndependentReparametrizationSampler(128, model, qmc=True)
    assert si.sample(x[:, :1, :]).shape == (5, 128, 1, 1)


def test_set_seed_makes_unseeded_draws_reproducible():
    """The analogue of tf.random.set_seed: un-seeded draws (space samples, eps, trajectories) repeat after
    set_seed and differ without it; an explicit seed argument always wins."""
    import trieste_amd
    from trieste_amd.rng import make_rng

    box = Box([0.0, 0.0], [1.0, 1.0])
    trieste_amd.set_seed(5)
    a, e1 = box.sample(7), make_rng().standard_normal(3)
    trieste_amd.set_seed(5)
    b, e2 = box.sample(7), make_rng().standard_normal(3)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(e1, e2)
    assert not np.array_equal(box.sample(7), b)  # the sequence moves on
    np.testing.assert_array_equal(box.sample(7, seed=3), box.sample(7, seed=3))
    trieste_amd.set_seed(None)
    assert not np.array_equal(box.sample(7), box.sample(7))


# ---- round-2 regressions (advisor findings) ---------------------------------------------------------------------
def test_history_records_hold_per_step_model_copies():
    """track_state=True: every Record owns deep copies of the models (reference bayesian_optimizer.py:745-760), so
    history[i].model still is the model as it was BEFORE step i + 1, not the final one."""
    model, data = _model(n=8)
    box = Box([0.0, 0.0], [1.0, 1.0])
    bo = BayesianOptimizer(lambda x: Dataset(x, OBJ.scaled_branin(x)), box)
    rule = EfficientGlobalOptimization(optimizer=generate_random_search_optimizer(200, seed=1, on_device=False))
    res = bo.optimize(3, data, model, rule, fit_model=True, fit_initial_model=False)
    final = res.final_result.unwrap()
    sizes = [len(rec.models[OBJECTIVE].get_internal_data()) for rec in res.history]
    assert sizes == [8, 9, 10] and len(final.models[OBJECTIVE].get_internal_data()) == 11
    assert all(rec.models[OBJECTIVE] is not final.models[OBJECTIVE] for rec in res.history)
    # a record holds the GPR record only: no engine (= no device memory) until the copy is first used
    assert all("_engine" not in rec.models[OBJECTIVE].__dict__ for rec in res.history)
    assert all(rec.models[OBJECTIVE].engine is not final.models[OBJECTIVE].engine for rec in res.history)
    assert [rec.models[OBJECTIVE].engine.N for rec in res.history] == [8, 9, 10]


def test_model_copies_are_lazy_and_keep_their_placement(monkeypatch):
    """copy.deepcopy(model): host record only; the engine is rebuilt (and refactorised) on first use with the SAME
    placement -- a devices=[...] model restored from the history is still sharded (round-2 advisor finding)."""
    import copy

    import trieste_amd.group as G
    from tests.fakes import FakeGroup

    monkeypatch.setattr(G, "GPEngineGroup", FakeGroup)
    model, data = _model(n=9)
    ref_mean = model.predict(data.query_points[:4])[0]
    version = model.data_version
    twin = copy.deepcopy(model)
    assert "_engine" not in twin.__dict__ and "_group" not in twin.__dict__
    np.testing.assert_allclose(twin.predict(data.query_points[:4])[0], ref_mean, rtol=1e-12)
    assert twin.engine is not model.engine and twin.group is None and twin.data_version == version
    x_new = np.array([[0.3, 0.3]])
    twin.update(data + Dataset(x_new, OBJ.scaled_branin(x_new)))
    assert twin.engine.N == 10 and model.engine.N == 9
    gpr = M.build_gpr(data, Box([0.0, 0.0], [1.0, 1.0]), likelihood_variance=1e-3)
    multi = M.GaussianProcessRegression(gpr, devices=[0, 1])
    mt = copy.deepcopy(multi)
    assert "_group" not in mt.__dict__
    assert mt.group is not None and mt.group is not multi.group and len(mt.group.members) == 2
    assert all(m.N == 9 for m in mt.group.members) and mt.engine is mt.group.primary


def test_fit_model_false_leaves_the_models_alone():
    """BayesianOptimizer fit_model=False: neither optimize nor update (reference bayesian_optimizer.py:828-834).
    Ask-Tell: `fit_model` is the constructor's INITIAL fit only, `tell` always updates
    (ask_tell_optimization.py:333-340, 716-718, 744-746); AskTellOptimizerNoTraining never touches the models
    (:749-757)."""
    model, data = _model(n=8)
    box = Box([0.0, 0.0], [1.0, 1.0])
    rule = EfficientGlobalOptimization(optimizer=generate_random_search_optimizer(100, seed=1, on_device=False))
    bo = BayesianOptimizer(lambda x: Dataset(x, OBJ.scaled_branin(x)), box)
    res = bo.optimize(2, data, model, rule, fit_model=False, track_state=False)
    assert len(res.final_result.unwrap().datasets[OBJECTIVE]) == 10 and model.engine.N == 8
    at = AskTellOptimizer(box, data, model, rule, fit_model=False)
    assert model.engine.N == 8  # no initial fit ...
    q = at.ask()
    at.tell(Dataset(q, OBJ.scaled_branin(q)))
    assert model.engine.N == 9  # ... but a pre-trained model handed over this way still sees the new observations
    q2 = at.ask()
    assert not np.allclose(q, q2)  # (it would keep proposing from stale data otherwise)
    from trieste_amd.ask_tell_optimization import AskTellOptimizerNoTraining

    model2, _ = _model(n=8)
    nt = AskTellOptimizerNoTraining(box, data, model2, rule)
    q = nt.ask()
    nt.tell(Dataset(q, OBJ.scaled_branin(q)))
    assert model2.engine.N == 8 and len(nt.dataset) == 9


def test_split_wrapper_keeps_the_fused_api_and_rejects_zero_vectorization():
    model, data = _model(n=10)
    fn = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    wrapped = split_acquisition_function(fn, 7)
    for attr in ("argmax", "top_k", "value_and_gradient", "_engine"):
        assert getattr(wrapped, attr) == getattr(fn, attr) or getattr(wrapped, attr) is getattr(fn, attr)
    with pytest.raises(ValueError):
        optimize_discrete(DiscreteSearchSpace(np.zeros((3, 2))), (fn, 0))


def test_a_failed_fit_does_not_poison_the_append_path():
    """If `optimize` dies after trying other hyper-parameters the engine is restored (or marked out of sync), so the
    next update never extends a factor built with the wrong kernel."""
    model, data = _model(n=9)
    ref_mean, _ = model.predict(data.query_points[:3])
    calls = {"n": 0}
    orig = model._loss_at

    def flaky(*a, **k):
        calls["n"] += 1
        out = orig(*a, **k)
        if calls["n"] >= 3:
            raise KeyboardInterrupt  # not an ArithmeticError: escapes scipy
        return out

    model._loss_at = flaky
    model._num_kernel_samples = 0
    with pytest.raises(KeyboardInterrupt):
        model.optimize(data)
    model._loss_at = orig
    assert model._in_sync
    np.testing.assert_allclose(model.predict(data.query_points[:3])[0], ref_mean, rtol=1e-12)
    x_new = np.array([[0.3, 0.3]])
    model.update(data + Dataset(x_new, OBJ.scaled_branin(x_new)))
    fresh, _ = _model(n=9)
    fresh.update(data + Dataset(x_new, OBJ.scaled_branin(x_new)))
    np.testing.assert_allclose(model.predict(x_new)[0], fresh.predict(x_new)[0], rtol=1e-9)


def test_model_with_devices_shards_the_fused_sweeps(monkeypatch):
    """GaussianProcessRegression(devices=[...]): one process, replicated updates, the fused arg-max / top-k of the plain
    posterior tails sharded over the group -- same winners as the single-device model, and the loop runs once."""
    import trieste_amd.group as G
    from tests.fakes import FakeGroup

    monkeypatch.setattr(G, "GPEngineGroup", FakeGroup)
    rng = np.random.default_rng(0)
    x = rng.uniform(size=(12, 2))
    data = Dataset(x, OBJ.scaled_branin(x))
    box = Box([0.0, 0.0], [1.0, 1.0])
    single = M.GaussianProcessRegression(M.build_gpr(data, box, likelihood_variance=1e-3))
    multi = M.GaussianProcessRegression(M.build_gpr(data, box, likelihood_variance=1e-3), devices=[0, 1, 2])
    assert multi.group is not None and len(multi.group.members) == 3 and single.group is None
    fs = ExpectedImprovement().prepare_acquisition_function(single, dataset=data)
    fm = ExpectedImprovement().prepare_acquisition_function(multi, dataset=data)
    pts = rng.uniform(size=(1001, 2))
    pts[900] = pts[17]
    a, b = fs.argmax(pts), fm.argmax(pts)
    assert (a[0], a[1]) == (b[0], b[1]) and np.array_equal(a[2], b[2])
    for u, v in zip(fs.top_k(pts, 7), fm.top_k(pts, 7)):
        np.testing.assert_array_equal(u, v)
    # device-side candidate generation: ONE logical Philox sample whatever the number of devices
    opt = generate_random_search_optimizer(3000, seed=11)
    np.testing.assert_array_equal(opt(box, fs), opt(box, fm))
    # updates are replicated (rank-k append on every member); the BO loop calls the observer once per step
    calls = []

    def observer(q):
        calls.append(len(q))
        return Dataset(q, OBJ.scaled_branin(q))

    res = BayesianOptimizer(observer, box).optimize(2, data, multi, EfficientGlobalOptimization(optimizer=opt),
                                                    fit_model=True, fit_initial_model=False, track_state=False)
    assert calls == [1, 1] and all(m.N == 14 for m in multi.group.members)
    assert len(res.final_result.unwrap().datasets[OBJECTIVE]) == 14
