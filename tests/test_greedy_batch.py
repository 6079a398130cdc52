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
