"""CPU tests of the entropy-search builders (MinValueEntropySearch, GIBBON) and the Gumbel sampler, written after
the reference's tests/unit/acquisition/function/test_entropy.py and tests/unit/acquisition/test_sampler.py
(cited per test).  The engine is replaced at its boundary by tests/fakes.py::FakeEngine (oracle-backed)."""
import numpy as np
import pytest

import trieste_amd.models as M
from oracle import gp_oracle as O
from tests.fakes import FakeEngine
from trieste_amd import objectives as OBJ
from trieste_amd.extras import (GIBBON, EfficientGlobalOptimization, ExactThompsonSampler, GibbonAcquisition,
                                     GumbelSampler, LocalPenalization, MinValueEntropySearch,
                                     ThompsonSamplerFromTrajectory, generate_continuous_optimizer,
                                     gibbon_quality_term, gibbon_repulsion_term, min_value_entropy_search)
from trieste_amd.data import Dataset
from trieste_amd.space import Box


@pytest.fixture(autouse=True)
def fake_engine(monkeypatch):
    monkeypatch.setattr(M, "GPEngine", FakeEngine)


def _model(n=14, d=2, noise=1e-2, seed=0, objective=OBJ.scaled_branin):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, d))
    data = Dataset(x, objective(x))
    gpr = M.build_gpr(data, Box([0.0] * d, [1.0] * d), likelihood_variance=noise)
    return M.GaussianProcessRegression(gpr), data


def _grid(n=11, lo=0.0, hi=1.0):
    r = np.linspace(lo, hi, n)
    return np.stack(np.meshgrid(r, r, indexing="ij"), axis=-1).reshape(-1, 2)


SPACE = Box([0, 0], [1, 1])


# ---- samplers (reference tests/unit/acquisition/test_sampler.py:35-78) -----------------------------------
def test_gumbel_sampler_errors_shapes_and_samples_are_minima():
    model, data = _model()
    with pytest.raises(ValueError):
        GumbelSampler(sample_min_value=False)
    for size in (0, -2):
        with pytest.raises(ValueError):
            GumbelSampler(True).sample(model, size, np.zeros((10, 2)))
    for shape in ((), (1,), (2,), (1, 2, 3)):
        with pytest.raises(ValueError):
            GumbelSampler(True).sample(model, 1, np.zeros(shape))
    at = np.concatenate([data.query_points, SPACE.sample(100, seed=1)])
    for size in (10, 100):
        assert GumbelSampler(True).sample(model, size, at).shape == (size, 1)
    samples = GumbelSampler(True).sample(model, 5, at)
    fmean, _ = model.predict(data.query_points)
    assert samples.max() < fmean.min() + 3 * np.sqrt(model.get_observation_noise())  # :61-78 (noisy observations here)
    # the restated algorithm, given the same uniform draws
    import trieste_amd

    trieste_amd.set_seed(77)
    got = GumbelSampler(True).sample(model, 6, at)
    trieste_amd.set_seed(77)
    from trieste_amd.rng import make_rng

    u = make_rng().uniform(size=6)
    ym, yv = model.predict_y(at)
    np.testing.assert_allclose(got, O.gumbel_min_value_samples(ym[:, 0], np.sqrt(yv[:, 0]), u), rtol=1e-9)


# ---- MinValueEntropySearch (reference test_entropy.py:57-268) -------------------------------------------
def test_mes_builder_errors_and_default_sampler():
    model, data = _model()
    with pytest.raises(ValueError):  # :57-70
        MinValueEntropySearch(SPACE).prepare_acquisition_function(model, dataset=Dataset(np.zeros((0, 2)), np.ones((0, 1))))
    with pytest.raises(ValueError):
        MinValueEntropySearch(SPACE).prepare_acquisition_function(model)
    for param in (-2, 0):  # :73-79
        with pytest.raises(ValueError):
            MinValueEntropySearch(SPACE, num_samples=param)
        with pytest.raises(ValueError):
            MinValueEntropySearch(SPACE, grid_size=param)
    for sampler in (ExactThompsonSampler(sample_min_value=False), ThompsonSamplerFromTrajectory(sample_min_value=False)):
        with pytest.raises(ValueError):  # :82-94
            MinValueEntropySearch(SPACE, min_value_sampler=sampler)
    builder = MinValueEntropySearch(SPACE)  # :97-101
    assert isinstance(builder._min_value_sampler, ExactThompsonSampler) and builder._min_value_sampler.sample_min_value
    for sampler in (ExactThompsonSampler(True), GumbelSampler(True), ThompsonSamplerFromTrajectory(True)):  # :104-115
        assert MinValueEntropySearch(SPACE, min_value_sampler=sampler)._min_value_sampler is sampler


@pytest.mark.parametrize("sampler", [ExactThompsonSampler(True), GumbelSampler(True), ThompsonSamplerFromTrajectory(True)],
                         ids=["exact", "gumbel", "trajectory"])
def test_mes_builder_builds_and_updates_min_value_samples(sampler):
    model, data = _model()  # :130-190
    builder = MinValueEntropySearch(SPACE, num_samples=7, grid_size=60, min_value_sampler=sampler)
    acq = builder.prepare_acquisition_function(model, dataset=data)
    assert isinstance(acq, min_value_entropy_search) and acq.samples.shape == (7, 1)
    fmean, _ = model.predict(data.query_points)
    assert np.all(acq.samples < fmean.min() + 0.5)  # samples of the minimum value
    before = acq.samples.copy()
    xs = _grid()
    v0 = np.array(acq(xs[:, None, :]))
    assert v0.shape == (121, 1) and np.all(np.isfinite(v0)) and np.all(v0 >= -1e-12)
    up = builder.update_acquisition_function(acq, model, dataset=data)
    assert up is acq and up.samples.shape == (7, 1) and not np.array_equal(up.samples, before)
    with pytest.raises(ValueError):
        builder.update_acquisition_function(lambda x: x, model, dataset=data)


def test_mes_function_shape_errors_and_oracle_values():
    model, data = _model()
    for samples in (np.array([]), np.array([[[]]])):  # :223-228
        with pytest.raises(ValueError):
            min_value_entropy_search(model, samples)
    acq = min_value_entropy_search(model, np.array([[1.0], [2.0]]))
    for at in (np.array([[0.0, 0.0], [1.0, 1.0]]), np.zeros((1, 2, 2))):  # :231-236
        with pytest.raises(ValueError):
            acq(at)
    xs = _grid(5)
    assert acq(xs[:, None, :]).shape == (25, 1)  # :239-244
    st = model.engine.state
    smp = np.array([[-0.2], [0.05], [-1.0]])
    acq.update(smp)
    m, v = O.predict(st, xs)
    np.testing.assert_allclose(acq(xs[:, None, :])[:, 0], O.min_value_entropy_search(m, v, smp[:, 0]), rtol=1e-12)
    # a [1, S] sample tensor is S samples too (tf.squeeze in the reference)
    acq2 = min_value_entropy_search(model, smp.T)
    np.testing.assert_array_equal(acq2(xs[:, None, :]), acq(xs[:, None, :]))
    # fused entry points agree with the values
    vals = acq(xs[:, None, :])[:, 0]
    v_, i_, x_ = acq.argmax(xs)
    assert i_ == int(np.argmax(vals)) and v_ == vals[i_]
    val, grad = acq.value_and_gradient(xs[:6] + 0.013)
    h = 1e-6
    num = np.stack([(acq((xs[:6] + 0.013 + h * e)[:, None, :]) - acq((xs[:6] + 0.013 - h * e)[:, None, :]))[:, 0] / (2 * h)
                    for e in np.eye(2)], axis=1)
    np.testing.assert_allclose(grad, num, rtol=2e-5, atol=1e-7 * np.abs(num).max())


def test_mes_chooses_same_as_probability_of_improvement():
    """One sample: MES is monotone in gamma, like PI with that threshold (Wang & Jegelka 2017; :247-268)."""
    model, data = _model(n=20)
    xs = _grid()
    sample = np.array([[float(model.engine.eta()) - 0.1]])
    mes = min_value_entropy_search(model, sample)(xs[:, None, :])[:, 0]
    mean, var = model.predict(xs)
    gamma = (sample[0, 0] - mean[:, 0]) / np.sqrt(var[:, 0])
    assert int(np.argmax(mes)) == int(np.argmax(O.normal_cdf(gamma)))


# ---- GIBBON (reference test_entropy.py:271-527) ---------------------------------------------------------
def test_gibbon_builder_errors_and_default_sampler():
    model, data = _model()
    with pytest.raises(ValueError):  # :271-284
        GIBBON(SPACE).prepare_acquisition_function(model, dataset=Dataset(np.zeros((0, 2)), np.ones((0, 1))))
    with pytest.raises(ValueError):
        GIBBON(SPACE).prepare_acquisition_function(model)
    for param in (-2, 0):  # :287-293
        with pytest.raises(ValueError):
            GIBBON(SPACE, num_samples=param)
        with pytest.raises(ValueError):
            GIBBON(SPACE, grid_size=param)
    with pytest.raises(ValueError):  # :296-308
        GIBBON(SPACE, min_value_sampler=ExactThompsonSampler(sample_min_value=False))
    assert isinstance(GIBBON(SPACE)._min_value_sampler, ExactThompsonSampler)  # :311-315

    class NoCovariance:
        pass

    with pytest.raises(NotImplementedError):
        GIBBON(SPACE).prepare_acquisition_function(NoCovariance(), dataset=data)
    for pending in (np.array([0.0]), np.zeros((1, 2, 2))):  # :424-434
        with pytest.raises(ValueError):
            GIBBON(SPACE, grid_size=30).prepare_acquisition_function(model, data, pending)
    for samples in (np.array([]), np.array([[[]]])):  # :343-349
        with pytest.raises(ValueError):
            gibbon_quality_term(model, samples)
    q = gibbon_quality_term(model, np.array([[1.0], [2.0]]))
    for at in (np.array([[0.0, 0.0], [1.0, 1.0]]), np.zeros((1, 2, 2))):  # :352-358
        with pytest.raises(ValueError):
            q(at)
    assert q(_grid(5)[:, None, :]).shape == (25, 1)  # :361-367


def test_gibbon_builder_builds_updates_and_switches_between_quality_and_batch_form():
    model, data = _model()  # :370-421
    builder = GIBBON(SPACE, num_samples=6, grid_size=50, min_value_sampler=GumbelSampler(True))
    acq = builder.prepare_acquisition_function(model, dataset=data)
    assert isinstance(acq, gibbon_quality_term) and acq.samples.shape == (6, 1)
    xs = _grid()
    st = model.engine.state
    m, v = O.predict(st, xs)
    np.testing.assert_allclose(acq(xs[:, None, :])[:, 0], O.gibbon_quality_term(m, v, acq.samples[:, 0], st.noise),
                               rtol=1e-12)
    pending = np.array([[0.3, 0.4], [0.8, 0.2]])
    batch = builder.update_acquisition_function(acq, model, data, pending[:1], new_optimization_step=False)
    assert isinstance(batch, GibbonAcquisition)
    batch2 = builder.update_acquisition_function(batch, model, data, pending, new_optimization_step=False)
    assert batch2 is batch  # in-place updates
    assert builder._diversity_term.conditioned_engine.N == len(data) + 2
    expected = (O.gibbon_quality_term(m, v, builder._quality_term.samples[:, 0], st.noise)
                + O.gibbon_repulsion_term(st, xs, pending, True))
    np.testing.assert_allclose(batch(xs[:, None, :])[:, 0], expected, rtol=1e-7, atol=1e-10)
    before = builder._quality_term.samples.copy()
    again = builder.update_acquisition_function(batch, model, data, None, new_optimization_step=True)
    assert again is acq and not np.array_equal(acq.samples, before)


def test_gibbon_chooses_same_as_min_value_entropy_search():
    model, data = _model(noise=1e-8)  # :466-481: one sample, negligible noise
    xs = _grid()
    sample = np.array([[float(model.engine.eta()) - 0.05]])
    mes = min_value_entropy_search(model, sample)(xs[:, None, :])
    gib = gibbon_quality_term(model, sample)(xs[:, None, :])
    assert int(np.argmax(mes)) == int(np.argmax(gib))


@pytest.mark.parametrize("rescaled_repulsion", [True, False])
@pytest.mark.parametrize("noise_variance", [0.1, 1e-8])
def test_batch_gibbon_is_sum_of_individual_gibbons_and_repulsion_term(rescaled_repulsion, noise_variance):
    """:484-527: the repulsion term against log-determinants of the joint predictive covariance."""
    model, data = _model(noise=noise_variance)
    xs = _grid(4)
    pending = np.array([[0.11, 0.51], [0.21, 0.31], [0.41, 0.91]])
    samples = np.array([[-0.1, 0.1]])
    quality = gibbon_quality_term(model, samples)
    repulsion = gibbon_repulsion_term(model, pending, rescaled_repulsion=rescaled_repulsion)
    fused = GibbonAcquisition(quality, repulsion)
    calculated = np.array(quality(xs[:, None, :])) + np.array(repulsion(xs[:, None, :]))
    np.testing.assert_allclose(fused(xs[:, None, :]), calculated, rtol=1e-9, atol=1e-12)
    _, pending_var = model.predict_joint(pending)
    pending_var = pending_var[0] + noise_variance * np.eye(3)
    for i in range(len(xs)):
        _, A = model.predict_joint(np.concatenate([xs[i:i + 1], pending], axis=0))
        A = A[0] + noise_variance * np.eye(4)
        rep = np.linalg.slogdet(A)[1] - np.log(A[0, 0]) - np.linalg.slogdet(pending_var)[1]
        if rescaled_repulsion:
            rep *= (1 / 3) ** 2
        np.testing.assert_allclose(calculated[i, 0], 0.5 * rep + quality(xs[i:i + 1, None, :])[0, 0], rtol=1e-5, atol=1e-6)
    # fused entry points; the engine state never leaks out of a call
    for name in ("argmax", "top_k", "value_and_gradient", "_engine"):
        assert hasattr(fused, name)
    vals = fused(xs[:, None, :])[:, 0]
    v_, i_, _ = fused.argmax(xs)
    assert i_ == int(np.argmax(vals)) and v_ == vals[i_]
    pts = xs[:5] + 0.017
    val, grad = fused.value_and_gradient(pts)
    h = 1e-6
    num = np.stack([(fused((pts + h * e)[:, None, :]) - fused((pts - h * e)[:, None, :]))[:, 0] / (2 * h)
                    for e in np.eye(2)], axis=1)
    np.testing.assert_allclose(val, fused(pts[:, None, :])[:, 0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(grad, num, rtol=1e-4, atol=1e-6 * np.abs(num).max())
    assert model.engine._rep is None
    np.testing.assert_allclose(quality(xs[:, None, :]), calculated - np.array(repulsion(xs[:, None, :])), rtol=1e-9,
                               atol=1e-12)
    generic = GibbonAcquisition(lambda x: np.full(x.shape[:-2] + (1,), 2.0), lambda x: np.full(x.shape[:-2] + (1,), -0.5))
    assert not hasattr(generic, "argmax")
    np.testing.assert_allclose(generic(xs[:, None, :]), 1.5)


def test_ego_with_entropy_builders():
    model, data = _model(n=16)
    opt = generate_continuous_optimizer(num_initial_samples=400, num_optimization_runs=3)
    pt = EfficientGlobalOptimization(MinValueEntropySearch(SPACE, grid_size=100), optimizer=opt).acquire_single(SPACE, model, data)
    assert pt.shape == (1, 2)
    rule = EfficientGlobalOptimization(GIBBON(SPACE, grid_size=100), optimizer=opt, num_query_points=3)
    pts = rule.acquire_single(SPACE, model, data)
    assert pts.shape == (3, 2) and np.all((pts >= 0) & (pts <= 1))
    dist = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1) + np.eye(3)
    assert dist.min() > 1e-3  # the repulsion term keeps the batch apart
    assert model.engine.N == 16
    pts2 = rule.acquire_single(SPACE, model, data)
    assert pts2.shape == (3, 2)
    # MES as the base of local penalization (the reference's second supported base, greedy_batch.py:86-91)
    lp = EfficientGlobalOptimization(LocalPenalization(SPACE, num_samples=100,
                                                       base_acquisition_function_builder=MinValueEntropySearch(SPACE, grid_size=100)),
                                     optimizer=opt, num_query_points=2)
    assert lp.acquire_single(SPACE, model, data).shape == (2, 2)
