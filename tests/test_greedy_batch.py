"""CPU tests of the greedy-batch builders (LocalPenalization, Fantasizer), written after the reference's
tests/unit/acquisition/function/test_greedy_batch.py (cited per test).  The engine is replaced at its
boundary by tests/fakes.py::FakeEngine (oracle-backed): host logic only, no HIP compute."""
import numpy as np
import pytest

import trieste_amd.models as M
from oracle import gp_oracle as O
from tests.fakes import FakeEngine
from trieste_amd import objectives as OBJ
from trieste_amd.acquisition import (EfficientGlobalOptimization, ExpectedImprovement, Fantasizer, LocalPenalization,
                                     NegativeLowerConfidenceBound, PenalizedAcquisition, expected_improvement,
                                     generate_continuous_optimizer, generate_random_search_optimizer,
                                     hard_local_penalizer, soft_local_penalizer)
from trieste_amd.acquisition.greedy_batch import _generate_fantasized_data
from trieste_amd.data import OBJECTIVE, Dataset
from trieste_amd.space import Box


@pytest.fixture(autouse=True)
def fake_engine(monkeypatch):
    monkeypatch.setattr(M, "GPEngine", FakeEngine)


def _model(n=12, d=2, noise=1e-3, seed=0, objective=OBJ.scaled_branin):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, d))
    data = Dataset(x, objective(x))
    gpr = M.build_gpr(data, Box([0.0] * d, [1.0] * d), likelihood_variance=noise)
    return M.GaussianProcessRegression(gpr), data


def _grid(n=11):
    r = np.linspace(0.0, 1.0, n)
    return np.stack(np.meshgrid(r, r, indexing="ij"), axis=-1).reshape(-1, 2)


# ---- LocalPenalization (reference test_greedy_batch.py:53-183) ------------------------------------------
def test_local_penalization_raises_for_empty_data_and_invalid_num_samples():
    model, _ = _model()
    space = Box([0, 0], [1, 1])
    with pytest.raises(ValueError):  # :53-64
        LocalPenalization(space).prepare_acquisition_function(model, dataset=Dataset(np.zeros((0, 2)), np.ones((0, 1))))
    with pytest.raises(ValueError):
        LocalPenalization(space).prepare_acquisition_function(model)
    with pytest.raises(ValueError):  # :67-70
        LocalPenalization(space, num_samples=-5)


@pytest.mark.parametrize("pending_points", [np.array([0.0]), np.array([[[0.0, 0.0], [1.0, 1.0]]])])
def test_local_penalization_raises_for_invalid_pending_points_shape(pending_points):
    model, data = _model()  # :73-82
    with pytest.raises(ValueError):
        LocalPenalization(Box([0, 0], [1, 1])).prepare_acquisition_function(model, data, pending_points)


def test_local_penalization_without_pending_points_is_the_base_acquisition():
    model, data = _model()  # :85-114
    lp = LocalPenalization(Box([0, 0], [1, 1])).prepare_acquisition_function(model, data, None)
    xs = _grid()
    # the builder reuses its own eta estimate (min mean over data + samples) for EI (:233-236)
    assert isinstance(lp, expected_improvement)
    np.testing.assert_array_equal(lp(xs[:, None, :]), expected_improvement(model, lp.eta)(xs[:, None, :]))
    assert lp.eta <= ExpectedImprovement().prepare_acquisition_function(model, data).eta + 1e-12


@pytest.mark.parametrize("penalizer", [soft_local_penalizer, hard_local_penalizer])
def test_local_penalization_combines_base_and_penalization_correctly(penalizer):
    model, data = _model()  # :117-157
    pending = np.array([[0.2, 0.3], [0.7, 0.6]])
    builder = LocalPenalization(Box([0, 0], [1, 1]), penalizer=penalizer)
    lp = builder.prepare_acquisition_function(model, data, None)
    lp = builder.update_acquisition_function(lp, model, data, pending[:1], False)
    up = builder.update_acquisition_function(lp, model, data, pending, False)
    assert up is lp  # in-place updates
    assert isinstance(lp, PenalizedAcquisition)
    best, lipschitz = builder._eta, builder._lipschitz_constant
    xs = _grid()
    base_values = expected_improvement(model, best)(xs[:, None, :])
    pen_values = penalizer(model, pending, lipschitz, best)(xs[:, None, :])
    with np.errstate(divide="ignore"):
        expected = np.exp(np.log(base_values) + np.log(pen_values))
    np.testing.assert_allclose(lp(xs[:, None, :]), expected, rtol=1e-13, atol=1e-300)
    # the penalizer parameters are the reference's (greedy_batch.py:287-300), the values the oracle's
    st = model.engine.state
    r, s = O.local_penalizer_parameters(st, pending, lipschitz, best)
    np.testing.assert_allclose(pen_values[:, 0], O.PENALIZERS[penalizer.kind](xs, pending, r, s), rtol=1e-13)
    # the estimate itself: max mean-gradient norm / min mean over data + samples; here only its consistency
    lip2, eta2 = O.lipschitz_estimate(st, data.query_points)
    assert lipschitz >= lip2 - 1e-12 and best <= eta2 + 1e-12
    # penalization pushes the maximiser away from the pending points
    assert np.all(lp(pending[:, None, :]) < base_values.max() * 1e-3 + expected_improvement(model, best)(pending[:, None, :]))
    # a new optimisation step without pending points hands back the (updated) base function
    base_again = builder.update_acquisition_function(lp, model, data, None, True)
    assert isinstance(base_again, expected_improvement)


@pytest.mark.parametrize("penalizer", [soft_local_penalizer, hard_local_penalizer])
def test_lipschitz_penalizers_raise_for_invalid_shapes(penalizer):
    model, _ = _model()  # :160-183
    lp = penalizer(model, np.zeros((1, 2)), 1.0, 0.0)
    for at in (np.array([[0.0, 0.0], [1.0, 1.0]]), np.zeros((1, 2, 2))):
        with pytest.raises(ValueError):
            lp(at)
    for pending in (np.array([0.0]), np.zeros((1, 2, 2))):
        with pytest.raises(ValueError):
            penalizer(model, pending, 1.0, 0.0)


def test_penalized_acquisition_exposes_the_fused_entry_points_only_when_it_runs_on_the_engine():
    model, data = _model()
    base = expected_improvement(model, model.engine.eta())
    pen = soft_local_penalizer(model, np.array([[0.4, 0.4]]), 2.0, model.engine.eta())
    fused = PenalizedAcquisition(base, pen)
    for name in ("argmax", "top_k", "value_and_gradient", "_engine"):
        assert hasattr(fused, name)
    xs = np.random.default_rng(0).uniform(size=(200, 2))
    vals = fused(xs[:, None, :])[:, 0]
    v, i, x = fused.argmax(xs)
    assert i == int(np.argmax(vals)) and v == vals[i] and np.array_equal(x, xs[i])
    tv, ti = fused.top_k(xs, 5)
    np.testing.assert_array_equal(ti, np.argsort(-vals, kind="stable")[:5])
    val, grad = fused.value_and_gradient(xs[:7])
    h = 1e-6
    num = np.stack([(fused((xs[:7] + h * e)[:, None, :]) - fused((xs[:7] - h * e)[:, None, :]))[:, 0] / (2 * h)
                    for e in np.eye(2)], axis=1)
    np.testing.assert_allclose(val, vals[:7], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(grad, num, rtol=1e-5, atol=1e-8 * np.abs(num).max())
    # the penalization never leaks out of a call
    np.testing.assert_array_equal(base(xs[:, None, :]), expected_improvement(model, base.eta)(xs[:, None, :]))
    assert model.engine._pen is None
    # foreign callables: combined from their values, no fused entry points
    generic = PenalizedAcquisition(lambda x: np.full(x.shape[:-2] + (1,), 2.0), lambda x: np.full(x.shape[:-2] + (1,), 0.25))
    assert not hasattr(generic, "argmax") and not hasattr(generic, "value_and_gradient")
    np.testing.assert_allclose(generic(xs[:, None, :]), 0.5)


@pytest.mark.parametrize("optimizer", [None, "random"])
def test_ego_with_local_penalization_returns_a_diverse_batch(optimizer):
    """rule.py:384-397: one optimisation per batch element, pending points growing."""
    model, data = _model(n=15)
    space = Box([0, 0], [1, 1])
    opt = generate_random_search_optimizer(2000, seed=3, on_device=False) if optimizer else \
        generate_continuous_optimizer(num_initial_samples=500, num_optimization_runs=4)
    rule = EfficientGlobalOptimization(LocalPenalization(space, num_samples=200), optimizer=opt, num_query_points=4)
    pts = rule.acquire_single(space, model, data)
    assert pts.shape == (4, 2) and np.all((pts >= 0) & (pts <= 1))
    dist = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1) + np.eye(4)
    assert dist.min() > 1e-3  # penalization keeps the batch apart
    # a second step reuses the builder's objects
    pts2 = rule.acquire_single(space, model, data)
    assert pts2.shape == (4, 2)


def test_local_penalization_on_a_multi_device_model_penalises_every_shard(monkeypatch):
    """A model built with devices=[...] shards the fused sweeps over its group; the penalization is handle state and
    has to reach EVERY member -- set on member 0 alone, the other shards are swept unpenalised and the greedy batch
    repeats its first point (round-2 advisor finding).  Same batch as the single-device model, point for point."""
    import trieste_amd.group as G
    from tests.fakes import FakeGroup

    monkeypatch.setattr(G, "GPEngineGroup", FakeGroup)
    rng = np.random.default_rng(0)
    x = rng.uniform(size=(15, 2))
    data = Dataset(x, OBJ.scaled_branin(x))

    class SeededBox(Box):  # the Lipschitz estimate samples the space: the same sample for both models
        def sample(self, num_samples, seed=None):
            return super().sample(num_samples, seed=123 if seed is None else seed)

    space = SeededBox([0, 0], [1, 1])
    single = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-3))
    multi = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-3), devices=[0, 1, 2])
    batches = []
    for model in (single, multi):
        opt = generate_random_search_optimizer(3000, seed=5, on_device=False)
        rule = EfficientGlobalOptimization(LocalPenalization(space, num_samples=200), optimizer=opt, num_query_points=4)
        batches.append(rule.acquire_single(space, model, data))
    np.testing.assert_array_equal(batches[0], batches[1])
    dist = np.linalg.norm(batches[1][:, None, :] - batches[1][None, :, :], axis=-1) + np.eye(4)
    assert dist.min() > 1e-3
    # the scope leaves no member penalised behind
    assert all(m._pen is None for m in multi.group.members)
    # and the fused arg-max of a penalised function is the single-device one on a table crossing the shard bounds
    lp1 = LocalPenalization(space, num_samples=200)
    lp3 = LocalPenalization(space, num_samples=200)
    pend = batches[0][:2]
    f1 = lp1.prepare_acquisition_function(single, data, pend)
    f3 = lp3.prepare_acquisition_function(multi, data, pend)
    pts = rng.uniform(size=(2999, 2))
    a, b = f1.argmax(pts), f3.argmax(pts)
    assert (a[0], a[1]) == (b[0], b[1])
    for u, v in zip(f1.top_k(pts, 5), f3.top_k(pts, 5)):
        np.testing.assert_array_equal(u, v)


# ---- Fantasizer (reference test_greedy_batch.py:187-296) ------------------------------------------------
def _sin_model():
    x = (np.arange(1, 6).reshape(-1, 1) / 5.0)
    y = 2.0 * np.sin(x / 3.0)  # fnc_2sin_x_over_3
    gpr = M.GPR((x, y), M.Matern52(1.0, 0.4), M.Constant(0.0), 1e-3)
    return M.GaussianProcessRegression(gpr), Dataset(x, y)


def test_fantasizer_raises_for_invalid_method_model_and_pending_points():
    with pytest.raises(ValueError):  # :187-189
        Fantasizer(ExpectedImprovement().using(OBJECTIVE), "notKB")
    model, data = _sin_model()

    class NotFantasizable:
        def predict(self, x):
            return np.zeros(x.shape[:-1] + (1,)), np.ones(x.shape[:-1] + (1,))

    with pytest.raises(NotImplementedError):  # :192-201
        Fantasizer().prepare_acquisition_function({OBJECTIVE: NotFantasizable()}, {OBJECTIVE: data}, np.zeros((3, 1)))
    for pending in (np.array([0.0]), np.array([[[0.0], [1.0]]])):  # :218-230
        with pytest.raises(ValueError):
            Fantasizer().prepare_acquisition_function({OBJECTIVE: model}, {OBJECTIVE: data}, pending)


def test_fantasize_with_kriging_believer_does_not_change_the_predictive_mean():
    model, data = _sin_model()  # :233-257 (NegativePredictiveMean = -LCB with beta = 0)
    x_test = (np.arange(1, 13).reshape(-1, 1) / 12.0)[:, None, :]
    pending = np.array([[0.51], [0.81]])
    builder = Fantasizer(NegativeLowerConfidenceBound(0.0))
    acq0 = builder.prepare_acquisition_function({OBJECTIVE: model}, {OBJECTIVE: data})
    v0 = np.array(acq0(x_test))
    acq1 = builder.prepare_acquisition_function({OBJECTIVE: model}, {OBJECTIVE: data}, pending)
    np.testing.assert_allclose(acq1(x_test), v0, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("method", ["KB", "sample"])
def test_fantasize_reduces_predictive_variance_and_updates_in_place(method):
    model, data = _sin_model()  # :260-296
    x_test = (np.arange(1, 13).reshape(-1, 1) / 12.0)
    pending = np.array([[0.51], [0.81]])
    builder = Fantasizer(ExpectedImprovement(), fantasize_method=method)
    models, datasets = {OBJECTIVE: model}, {OBJECTIVE: data}
    acq0 = builder.prepare_acquisition_function(models, datasets)
    acq1 = builder.update_acquisition_function(acq0, models, datasets, pending[:1])
    fm = builder._fantasized_models[OBJECTIVE]
    _, var0 = model.predict(x_test)
    _, var1 = fm.predict(x_test)
    assert np.all(var1 < var0)
    clones, appends = FakeEngine.cloned, FakeEngine.appended
    acq1_up = builder.update_acquisition_function(acq1, models, datasets, pending)
    assert acq1_up is acq1 and builder._fantasized_models[OBJECTIVE] is fm  # in-place updates
    _, var2 = fm.predict(x_test)
    assert np.all(var2 < var1 + 1e-15) and fm.engine.N == 7
    if method == "KB":  # the believer's earlier fantasies are unchanged: only the new row is appended
        assert FakeEngine.cloned == clones and FakeEngine.appended == appends + 1
    acq0_up = builder.update_acquisition_function(acq1, models, datasets)
    assert acq0_up is acq0
    # the base model never saw the fantasies
    assert model.engine.N == 5 and len(model.get_internal_data()) == 5


@pytest.mark.parametrize("method", ["KB", "sample"])
def test_fantasized_model_is_the_reference_conditional_posterior(method):
    """_fantasized_model.predict / predict_joint / predict_y are conditional_predict_f / _joint / _y of the base
    model with the fantasized data (greedy_batch.py:669-764)."""
    model, data = _model(n=14, noise=1e-2)
    pending = np.array([[0.3, 0.3], [0.8, 0.1], [0.5, 0.9]])
    fant = _generate_fantasized_data(method, model, pending)
    assert fant.query_points.shape == (3, 2) and fant.observations.shape == (3, 1)
    if method == "KB":
        np.testing.assert_array_equal(fant.observations, model.predict(pending)[0])
    fm = M.FantasizedGaussianProcessRegression(model, fant)
    xs = np.random.default_rng(5).uniform(size=(9, 2))
    cm, cv = model.conditional_predict_f(xs, fant)
    m, v = fm.predict(xs)
    np.testing.assert_allclose(m, cm, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(v, np.maximum(cv, 1e-12), rtol=1e-6, atol=1e-10)
    jm, jc = model.conditional_predict_joint(xs, fant)
    m2, c2 = fm.predict_joint(xs)
    np.testing.assert_allclose(m2, jm, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(c2, jc, rtol=1e-6, atol=1e-9)
    ym, yv = model.conditional_predict_y(xs, fant)
    np.testing.assert_allclose(fm.predict_y(xs)[1], np.maximum(cv, 1e-12) + model.get_observation_noise(), rtol=1e-6)
    assert fm.get_kernel() is model.get_kernel() and fm.get_observation_noise() == model.get_observation_noise()
    assert len(fm.get_internal_data()) == 17
    with pytest.raises(NotImplementedError):
        fm.update(data)
    with pytest.raises(NotImplementedError):  # leading dimensions of fantasized data: not on this engine
        fm.update_fantasized_data(Dataset(np.zeros((2, 3, 2)), np.zeros((2, 3, 1))))
    # a changed base model invalidates the incremental path: the clone is refreshed
    model.update(data + Dataset(np.array([[0.11, 0.12]]), np.array([[0.5]])))
    fm.update_fantasized_data(fant + Dataset(np.array([[0.6, 0.6]]), np.array([[0.1]])))
    assert fm.engine.N == 15 + 4
    np.testing.assert_allclose(fm.predict(xs)[0],
                               model.conditional_predict_f(xs, fant + Dataset(np.array([[0.6, 0.6]]), np.array([[0.1]])))[0],
                               rtol=1e-8, atol=1e-10)


def test_ego_with_fantasizer_returns_a_batch_and_leaves_the_model_alone():
    model, data = _model(n=15)
    space = Box([0, 0], [1, 1])
    rule = EfficientGlobalOptimization(Fantasizer(), num_query_points=3,
                                       optimizer=generate_continuous_optimizer(num_initial_samples=400,
                                                                               num_optimization_runs=4))
    pts = rule.acquire_single(space, model, data)
    assert pts.shape == (3, 2) and np.all((pts >= 0) & (pts <= 1))
    dist = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1) + np.eye(3)
    assert dist.min() > 1e-4  # EI at a believed point is ~0: the batch spreads
    assert model.engine.N == 15
    pts2 = rule.acquire_single(space, model, data)  # next step: same objects, fresh clone
    assert pts2.shape == (3, 2)


# ---- small siblings (reference tests/unit/acquisition/function/test_function.py) --------------------------
def test_negative_predictive_mean_probability_of_feasibility_and_predictive_variance():
    from scipy.stats import norm

    from trieste_amd.acquisition import (NegativePredictiveMean, PredictiveVariance, ProbabilityOfFeasibility,
                                         predictive_variance)

    model, data = _model()
    xs = _grid(6)
    mean, var = model.predict(xs)
    npm = NegativePredictiveMean().prepare_acquisition_function(model, data)
    np.testing.assert_allclose(npm(xs[:, None, :]), -mean, rtol=1e-12)
    with pytest.raises(ValueError):
        ProbabilityOfFeasibility(np.array([1.0, 2.0]))
    pof = ProbabilityOfFeasibility(0.3)
    assert pof.threshold == 0.3
    fn = pof.prepare_acquisition_function(model)
    np.testing.assert_allclose(fn(xs[:, None, :]), norm.cdf((0.3 - mean) / np.sqrt(var)), rtol=1e-10)
    assert pof.update_acquisition_function(fn, model) is fn
    pv = PredictiveVariance().prepare_acquisition_function(model)
    np.testing.assert_allclose(pv(xs[:, None, :]), var + 1e-6, rtol=1e-9)  # batch of one: the variance (+ jitter)
    batch = np.random.default_rng(0).uniform(size=(5, 3, 2))
    _, cov = model.predict_joint(batch)
    np.testing.assert_allclose(pv(batch), np.exp(np.linalg.slogdet(cov + 1e-6)[1]), rtol=1e-10)

    class NoJoint:
        pass

    with pytest.raises(NotImplementedError):
        predictive_variance(NoJoint(), 1e-6)


def test_make_positive_keeps_the_fused_entry_points_and_feeds_local_penalization():
    from trieste_amd.acquisition import MakePositive, NegativePredictiveMean

    model, data = _model()
    builder = MakePositive(NegativePredictiveMean())
    fn = builder.prepare_acquisition_function(model, data)
    xs = np.random.default_rng(2).uniform(size=(150, 2))
    base = -model.predict(xs)[0]
    vals = fn(xs[:, None, :])
    np.testing.assert_allclose(vals, np.log1p(np.exp(base)), rtol=1e-12)
    assert np.all(vals > 0)
    assert builder.update_acquisition_function(fn, model, data) is fn
    v, i, x = fn.argmax(xs)
    assert i == int(np.argmax(vals)) and np.isclose(v, vals[i, 0]) and np.array_equal(x, xs[i])
    tv, ti = fn.top_k(xs, 4)
    np.testing.assert_array_equal(ti, np.argsort(-vals[:, 0], kind="stable")[:4])
    val, grad = fn.value_and_gradient(xs[:5])
    h = 1e-6
    num = np.stack([(fn((xs[:5] + h * e)[:, None, :]) - fn((xs[:5] - h * e)[:, None, :]))[:, 0] / (2 * h) for e in np.eye(2)], axis=1)
    np.testing.assert_allclose(grad, num, rtol=1e-5, atol=1e-8)
    assert not hasattr(MakePositive(ExpectedImprovement()).prepare_acquisition_function(model, data), "nonexistent")
    # the reference's use: a strictly positive base for local penalization (greedy_batch.py:86-91)
    space = Box([0, 0], [1, 1])
    rule = EfficientGlobalOptimization(LocalPenalization(space, num_samples=100, base_acquisition_function_builder=builder),
                                       optimizer=generate_random_search_optimizer(500, seed=2, on_device=False),
                                       num_query_points=3)
    pts = rule.acquire_single(space, model, data)
    assert pts.shape == (3, 2)


def test_multiple_optimism_lower_confidence_bound():
    from scipy.stats import norm

    from trieste_amd.acquisition import MultipleOptimismNegativeLowerConfidenceBound, multiple_optimism_lower_confidence_bound

    model, data = _model()
    space = Box([0, 0], [1, 1])
    with pytest.raises(ValueError):
        multiple_optimism_lower_confidence_bound(model, 0)
    builder = MultipleOptimismNegativeLowerConfidenceBound(space)
    fn = builder.prepare_acquisition_function(model, data)
    assert builder.update_acquisition_function(fn, model, data) is fn
    with pytest.raises(ValueError):
        builder.update_acquisition_function(lambda x: x, model, data)
    B = 4
    x = np.random.default_rng(1).uniform(size=(30, B, 2))
    vals = fn(x)
    assert vals.shape == (30, B)
    betas = 5.0 * 2 * norm.ppf(0.5 + 0.5 * np.arange(1, B + 1) / (B + 1.0))
    mean, var = model.predict(x)
    np.testing.assert_allclose(vals, -mean[..., 0] + np.sqrt(var[..., 0]) * betas, rtol=1e-10)
    with pytest.raises(ValueError):  # fixed batch size
        fn(x[:, :2, :])
    val, grad = fn.value_and_gradient(x[:3])
    assert val.shape == (3, B) and grad.shape == (3, B, 2)
    np.testing.assert_allclose(val, vals[:3], rtol=1e-10)
    # a vectorized builder: EGO optimises the B columns independently
    rule = EfficientGlobalOptimization(MultipleOptimismNegativeLowerConfidenceBound(space), num_query_points=3,
                                       optimizer=generate_continuous_optimizer(num_initial_samples=200, num_optimization_runs=3))
    pts = rule.acquire_single(space, model, data)
    assert pts.shape == (3, 2) and np.all((pts >= 0) & (pts <= 1))


# ---- ExpectedConstrainedImprovement (reference test_function.py:928-1128) ---------------------------------
class _FnBuilder:
    """A constraint builder returning a fixed function of x [..., 1, D] -> [..., 1]."""

    def __new__(cls, fn):
        from trieste_amd.acquisition import AcquisitionFunctionBuilder

        class B(AcquisitionFunctionBuilder):
            def prepare_acquisition_function(self, models, datasets=None):
                return fn

        return B()


def test_expected_constrained_improvement():
    from trieste_amd.acquisition import ExpectedConstrainedImprovement, ProbabilityOfFeasibility

    FOO, CON = "foo", "constraint"
    model, data = _model(n=14)
    models, datasets = {FOO: model}, {FOO: data}
    certainty = _FnBuilder(lambda x: np.ones(np.asarray(x).shape[:-2] + (1,)))
    for bad in (np.array([0.5, 0.5]),):  # :928-931
        with pytest.raises(ValueError):
            ExpectedConstrainedImprovement(FOO, certainty, bad)
    for bad in (-0.1, 1.1):  # :934-938
        with pytest.raises(ValueError):
            ExpectedConstrainedImprovement(FOO, certainty, bad)
    with pytest.raises(ValueError):  # :1046-1073
        ExpectedConstrainedImprovement(FOO, certainty).prepare_acquisition_function(
            models, datasets={FOO: Dataset(np.zeros((0, 2)), np.zeros((0, 1)))})
    with pytest.raises(ValueError):
        ExpectedConstrainedImprovement(FOO, certainty).prepare_acquisition_function(models)
    # a certain constraint reproduces EI, also after an update (:954-982)
    builder = ExpectedConstrainedImprovement(FOO, certainty, 0)
    eci = builder.prepare_acquisition_function(models, datasets=datasets)
    ei = ExpectedImprovement().using(FOO).prepare_acquisition_function(models, datasets=datasets)
    at = np.random.default_rng(0).uniform(size=(7, 1, 2))
    np.testing.assert_allclose(eci(at), ei(at), rtol=1e-12)
    for a in (np.zeros((2, 2, 2)),):  # batch size must be one
        with pytest.raises(ValueError):
            eci(a)
    assert builder.update_acquisition_function(eci, models, datasets=datasets) is eci
    # improvement is relative to the best FEASIBLE point (:995-1019)
    half = _FnBuilder(lambda x: (np.asarray(x)[..., 0, :1] >= 0.5).astype(float))
    eci2 = ExpectedConstrainedImprovement(FOO, half).prepare_acquisition_function(models, datasets=datasets)
    feas = data.query_points[:, 0] >= 0.5
    eta = float(np.min(model.predict(data.query_points[feas])[0]))
    x = np.array([[[0.7, 0.3]]])
    np.testing.assert_allclose(eci2(x), expected_improvement(model, eta)(x), rtol=1e-12)
    assert float(eci2(np.array([[[0.2, 0.3]]]))[0, 0]) == 0.0  # infeasible candidate
    # no feasible point: the constraint function itself (:1076-1103)
    never = _FnBuilder(lambda x: np.zeros(np.asarray(x).shape[:-2] + (1,)))
    fn = ExpectedConstrainedImprovement(FOO, never).prepare_acquisition_function(models, datasets=datasets)
    np.testing.assert_array_equal(fn(at), np.zeros((7, 1)))
    # the bound is inclusive (:1106-1128)
    thr = float(1 / (1 + np.exp(-1.0)))
    sig = _FnBuilder(lambda x: 1 / (1 + np.exp(-np.asarray(x)[..., 0, :1] * 0 - 1.0)))  # pof == sigmoid(1) everywhere
    eci3 = ExpectedConstrainedImprovement(FOO, sig, min_feasibility_probability=thr).prepare_acquisition_function(
        models, datasets=datasets)
    np.testing.assert_allclose(eci3(at), ei(at) * thr, rtol=1e-12)
    # the real thing: a second GPR as the constraint model, PoF as the constraint, EGO with the gradient optimizer
    cx = np.random.default_rng(5).uniform(size=(12, 2))
    cdata = Dataset(cx, (cx[:, :1] - 0.5))  # feasible where x0 < 0.5 (values below the threshold 0)
    cmodel = M.GaussianProcessRegression(M.build_gpr(cdata, Box([0, 0], [1, 1]), likelihood_variance=1e-3))
    builder = ExpectedConstrainedImprovement(FOO, ProbabilityOfFeasibility(0.0).using(CON), 0.5)
    models2, datasets2 = {FOO: model, CON: cmodel}, {FOO: data, CON: cdata}
    fn = builder.prepare_acquisition_function(models2, datasets2)
    pts = np.random.default_rng(6).uniform(size=(6, 2))
    val, grad = fn.value_and_gradient(pts)
    np.testing.assert_allclose(val, fn(pts[:, None, :])[:, 0], rtol=1e-10, atol=1e-14)
    h = 1e-6
    num = np.stack([(fn((pts + h * e)[:, None, :]) - fn((pts - h * e)[:, None, :]))[:, 0] / (2 * h) for e in np.eye(2)], axis=1)
    np.testing.assert_allclose(grad, num, rtol=1e-5, atol=1e-8)
    space = Box([0, 0], [1, 1])
    rule = EfficientGlobalOptimization(builder, optimizer=generate_continuous_optimizer(num_initial_samples=300,
                                                                                         num_optimization_runs=3))
    pt = rule.acquire(space, models2, datasets2)
    assert pt.shape == (1, 2)
    assert float(ProbabilityOfFeasibility(0.0).prepare_acquisition_function(cmodel)(pt[:, None, :])[0, 0]) > 0.3


def test_reducers_sum_product_map():
    """reference tests/unit/acquisition/test_combination.py: constituent functions are prepared / updated
    individually, outputs reduced elementwise."""
    from trieste_amd.extras import (Map, NegativeLowerConfidenceBound, ProbabilityOfFeasibility, Product, Reducer,
                                         Sum)

    model, data = _model()
    models, datasets = {OBJECTIVE: model}, {OBJECTIVE: data}
    with pytest.raises(ValueError):
        Sum()
    ei = ExpectedImprovement().using(OBJECTIVE)
    lcb = NegativeLowerConfidenceBound(1.0).using(OBJECTIVE)
    pof = ProbabilityOfFeasibility(0.4).using(OBJECTIVE)
    xs = np.random.default_rng(3).uniform(size=(40, 1, 2))
    parts = [b.prepare_acquisition_function(models, datasets)(xs) for b in (ei, lcb, pof)]
    s = Sum(ei, lcb, pof)
    assert s.acquisitions == (ei, lcb, pof) and "Sum(" in repr(s)
    fs = s.prepare_acquisition_function(models, datasets)
    np.testing.assert_allclose(fs(xs), parts[0] + parts[1] + parts[2], rtol=1e-12)
    fp = Product(ei, pof).prepare_acquisition_function(models, datasets)
    np.testing.assert_allclose(fp(xs), parts[0] * parts[2], rtol=1e-12)
    fm = Map(lambda v: -2.0 * v, lcb).prepare_acquisition_function(models, datasets)
    np.testing.assert_allclose(fm(xs), -2.0 * parts[1], rtol=1e-12)
    # update re-uses the constituent functions (EI's eta is refreshed in place)
    more = data + Dataset(np.array([[0.5, 0.5]]), np.array([[-5.0]]))
    model.update(more)
    before = s.functions[0]
    fs2 = s.update_acquisition_function(fs, models, {OBJECTIVE: more})
    assert s.functions[0] is before
    ei2 = ExpectedImprovement().using(OBJECTIVE).prepare_acquisition_function(models, {OBJECTIVE: more})
    np.testing.assert_allclose(fs2(xs) - s.functions[1](xs) - s.functions[2](xs), ei2(xs), rtol=1e-9, atol=1e-12)
    # drives EGO through the generic optimizer path
    pt = EfficientGlobalOptimization(Product(ei, pof), optimizer=generate_random_search_optimizer(
        400, seed=1, on_device=False)).acquire(Box([0, 0], [1, 1]), models, datasets)
    assert pt.shape == (1, 2)
    assert issubclass(Sum, Reducer)
