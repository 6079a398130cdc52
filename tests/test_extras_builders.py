"""CPU tests of the OUT-OF-SCOPE builders frozen under trieste_amd/extras (builders.py, combination.py): moved out of
tests/test_greedy_batch.py / tests/test_host_logic.py in round 6 together with the classes.
The engine is replaced at its boundary by tests/fakes.py::FakeEngine (oracle-backed): host logic only, no HIP compute."""
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


def test_negative_predictive_mean_probability_of_feasibility_and_predictive_variance():
    from scipy.stats import norm

    from trieste_amd.extras import (NegativePredictiveMean, PredictiveVariance, ProbabilityOfFeasibility,
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
    from trieste_amd.extras import MakePositive, NegativePredictiveMean

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

    from trieste_amd.extras import MultipleOptimismNegativeLowerConfidenceBound, multiple_optimism_lower_confidence_bound

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
    from trieste_amd.extras import ExpectedConstrainedImprovement, ProbabilityOfFeasibility

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


def test_multiple_optimism_lcb_accepts_flat_points_and_single_query_point():
    from trieste_amd.extras import MultipleOptimismNegativeLowerConfidenceBound

    model, data = _model(n=10)
    box = Box([0.0, 0.0], [1.0, 1.0])
    fn = MultipleOptimismNegativeLowerConfidenceBound(box).prepare_acquisition_function(model, dataset=data)
    pts = np.random.default_rng(0).uniform(size=(5, 2))
    v2, g2 = fn.value_and_gradient(pts)                 # [P, D]: what batch-size-one optimizers pass
    v3, g3 = fn.value_and_gradient(pts[:, None, :])
    assert v2.shape == (5,) and g2.shape == (5, 2)
    np.testing.assert_array_equal(v2, v3[:, 0])
    np.testing.assert_array_equal(g2, g3[:, 0, :])
    rule = EfficientGlobalOptimization(MultipleOptimismNegativeLowerConfidenceBound(box), num_query_points=1)
    assert rule.acquire_single(box, model, dataset=data).shape == (1, 2)
