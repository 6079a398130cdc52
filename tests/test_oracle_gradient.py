"""CPU: the oracle's analytic acquisition gradient vs central finite differences of its own values."""
import numpy as np
import pytest

from oracle import gp_oracle as O


@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("acq", ["ei", "pi", "nlcb", "aei"])
def test_gradient_matches_finite_differences(kind, acq):
    rng = np.random.default_rng(0)
    X, Y = O.synthetic_problem(O.hartmann_6, 6, 40)
    ls = O.default_lengthscales(6) * np.linspace(0.8, 1.4, 6)
    st = O.gpr_update(kind, 1.3, ls, 1e-2, float(Y.mean()), X, Y)
    Xq = rng.uniform(size=(6, 6))
    par = O.eta_min_mean(st) if acq != "nlcb" else 1.96
    tails = {"ei": O.expected_improvement, "pi": O.probability_of_improvement,
             "nlcb": O.negative_lower_confidence_bound,
             "aei": lambda m, v, p: O.augmented_expected_improvement(m, v, p, st.noise)}

    def f(x):
        m, v = O.predict(st, x)
        return tails[acq](m, v, par)

    val, grad = O.acq_value_and_grad(st, acq, par, Xq)
    np.testing.assert_allclose(val, f(Xq), rtol=1e-10, atol=1e-14)
    h = 1e-6
    num = np.stack([(f(Xq + h * e) - f(Xq - h * e)) / (2 * h) for e in np.eye(6)], axis=1)
    np.testing.assert_allclose(grad, num, rtol=1e-6, atol=1e-7 * np.abs(num).max())


def test_clipped_variance_has_zero_variance_gradient():
    X, Y = O.synthetic_problem(O.branin, 2, 20)
    st = O.gpr_update("matern52", 1.0, O.default_lengthscales(2), 1e-14, 0.0, X, Y)
    val, grad = O.acq_value_and_grad(st, "nlcb", 2.0, X[:3])  # at training inputs: var clipped to 1e-12
    _, grad_mu = O.acq_value_and_grad(st, "nlcb", 0.0, X[:3])  # beta = 0: the pure mean gradient
    np.testing.assert_allclose(grad, grad_mu, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32", "matern52"])
def test_trajectory_gradient_matches_finite_differences(kind):
    rng = np.random.default_rng(1)
    d, N, F, B, P = 4, 30, 64, 3, 5
    X, Y = O.synthetic_problem(O.ackley, d, N)
    ls = O.default_lengthscales(d) * np.linspace(0.7, 1.2, d)
    st = O.gpr_update(kind, 0.9, ls, 1e-2, float(Y.mean()), X, Y)
    W, b = rng.standard_normal((F, d)), rng.uniform(0, 2 * np.pi, F)
    w, xi = rng.standard_normal((F, B)), rng.standard_normal((N, B))
    v = O.decoupled_weights(st, W, b, w, xi)
    Xq = rng.uniform(size=(P, B, d))
    val, grad = O.trajectory_value_and_grad(st, W, b, w, v, Xq)
    np.testing.assert_allclose(val, O.trajectory_eval(st, W, b, w, v, Xq), rtol=1e-12, atol=1e-12)
    h = 1e-6
    for c in range(d):
        e = np.zeros(d); e[c] = h
        num = (O.trajectory_eval(st, W, b, w, v, Xq + e) - O.trajectory_eval(st, W, b, w, v, Xq - e)) / (2 * h)
        np.testing.assert_allclose(grad[:, :, c], num, rtol=2e-6, atol=1e-7 * np.abs(num).max())


def test_conditional_predict_equals_refit_on_augmented_data():
    """models.py:355-484: conditioning on additional noisy observations == refitting the same
    hyper-parameters on the augmented data set (the identity the reference tests use,
    tests/unit/models/gpflow/test_models.py conditional_predict tests)."""
    rng = np.random.default_rng(2)
    d, N, n, M = 3, 25, 4, 7
    X, Y = O.synthetic_problem(O.ackley, d, N)
    ls = O.default_lengthscales(d)
    st = O.gpr_update("matern52", 1.1, ls, 1e-2, 0.3, X, Y)
    Xa, Ya = rng.uniform(size=(n, d)), rng.standard_normal(n)
    Xq = rng.uniform(size=(M, d))
    st2 = O.gpr_update("matern52", 1.1, ls, 1e-2, 0.3, np.concatenate([X, Xa]), np.concatenate([Y, Ya]))
    m_ref, c_ref = O.predict_joint(st2, Xq)
    m, c = O.conditional_predict_joint(st, Xq, Xa, Ya)
    np.testing.assert_allclose(m, m_ref, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(c, c_ref, rtol=1e-8, atol=1e-10)
    mf, vf = O.conditional_predict_f(st, Xq, Xa, Ya)
    np.testing.assert_allclose(mf, m_ref, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(vf, np.diag(c_ref), rtol=1e-8, atol=1e-10)


def test_rff_weight_posterior_design_and_gram_space_agree():
    """sampler.py:529-591: the two computation strategies describe the same Gaussian over theta -- equal
    posterior means, and equal covariances (compared through many draws mapped with the same eps is not
    possible: the Cholesky factors differ; compare mean and covariance directly)."""
    rng = np.random.default_rng(4)
    d, N, F = 3, 25, 40
    X, Y = O.synthetic_problem(O.ackley, d, N)
    st = O.gpr_update("rbf", 1.2, O.default_lengthscales(d), 5e-2, 0.1, X, Y)
    W, b = rng.standard_normal((F, d)), rng.uniform(0, 2 * np.pi, F)
    phi = O.rff_features(st, X, W, b)
    # design-space moments
    D = phi.T @ phi + st.noise * np.eye(F)
    mean_d = np.linalg.solve(D, phi.T @ st.err)
    cov_d = st.noise * np.linalg.inv(D)
    # gram-space moments
    G = phi @ phi.T + st.noise * np.eye(N)
    mean_g = phi.T @ np.linalg.solve(G, st.err)
    cov_g = np.eye(F) - phi.T @ np.linalg.solve(G, phi)
    np.testing.assert_allclose(mean_d, mean_g, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(cov_d, cov_g, rtol=1e-7, atol=1e-10)
    # the oracle picks gram space here (N <= F) and design space with fewer features; zero draws give the mean
    np.testing.assert_allclose(O.rff_theta(st, W, b, np.zeros((F, 1)))[:, 0], mean_g, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(O.rff_theta(st, W[:10], b[:10], np.zeros((10, 1)))[:, 0],
                               np.linalg.solve(phi[:, :10].T @ phi[:, :10] * (F / 10) + st.noise * np.eye(10),
                                               np.sqrt(F / 10) * phi[:, :10].T @ st.err), rtol=1e-8, atol=1e-10)
    # sample covariance of theta over many draws matches the analytic covariance
    th = O.rff_theta(st, W, b, rng.standard_normal((F, 20000)))
    np.testing.assert_allclose(np.cov(th), cov_g, atol=0.03)


@pytest.mark.parametrize("kind", ["soft", "hard"])
@pytest.mark.parametrize("acq", ["ei", "aei"])
def test_penalized_gradient_matches_finite_differences(kind, acq):
    """d/dx [a(x) prod_p phi_p(x)] (what autodiff through PenalizedAcquisition, greedy_batch.py:265-269, hands
    L-BFGS-B) vs central differences of the value; the Lipschitz estimate uses the same mean gradient."""
    rng = np.random.default_rng(3)
    d = 4
    X, Y = O.synthetic_problem(O.ackley, d, 30)
    st = O.gpr_update("matern52", 1.1, O.default_lengthscales(d), 1e-2, float(Y.mean()), X, Y)
    eta = O.eta_min_mean(st)
    pending = rng.uniform(size=(3, d))
    lip, eta_s = O.lipschitz_estimate(st, np.concatenate([X, rng.uniform(size=(50, d))]))
    assert lip > 0 and eta_s <= eta + 1e-12
    radius, scale = O.local_penalizer_parameters(st, pending, lip, eta_s)
    Xq = np.concatenate([rng.uniform(size=(5, d)), pending[:1] + 0.05])
    tails = {"ei": lambda m, v: O.expected_improvement(m, v, eta),
             "aei": lambda m, v: O.augmented_expected_improvement(m, v, eta, st.noise)}

    def f(x):
        m, v = O.predict(st, x)
        return tails[acq](m, v) * O.PENALIZERS[kind](x, pending, radius, scale)

    val, grad = O.penalized_value_and_grad(st, acq, eta, kind, pending, radius, scale, Xq)
    np.testing.assert_allclose(val, f(Xq), rtol=1e-10, atol=1e-16)
    h = 1e-6
    num = np.stack([(f(Xq + h * e) - f(Xq - h * e)) / (2 * h) for e in np.eye(d)], axis=1)
    np.testing.assert_allclose(grad, num, rtol=2e-6, atol=1e-7 * np.abs(num).max())
    # at a pending point itself the distance has no gradient: that term is dropped, the result stays finite
    _, g0 = O.penalized_value_and_grad(st, acq, eta, kind, pending, radius, scale, pending[:1])
    assert np.all(np.isfinite(g0))


@pytest.mark.parametrize("acq", ["mes", "gibbon"])
def test_entropy_gradient_matches_finite_differences(acq):
    """d/dx of min_value_entropy_search / GibbonAcquisition (entropy.py:195-214, 422-436, 479-500, 580-619) in
    analytic form vs central differences of the reference-form values, in the regime the acquisition is used in
    (samples a little below the best mean, candidates near the data)."""
    rng = np.random.default_rng(4)
    d = 3
    X, Y = O.synthetic_problem(O.hartmann_6, 6, 40)
    X, Y = X[:, :d], Y
    st = O.gpr_update("matern52", float(np.var(Y)), O.default_lengthscales(d), 1e-3, float(Y.mean()), X, Y)
    samples = O.eta_min_mean(st) - np.array([0.02, 0.1, 0.25, 0.4])
    pending = rng.uniform(size=(3, d))
    twin = O.fantasized_state(st, pending, np.zeros(3))
    Xq = np.clip(X[np.argsort(Y)[:6]] + 0.05 * rng.standard_normal((6, d)), 0, 1)

    def f(x):
        m, v = O.predict(st, x)
        if acq == "mes":
            return O.min_value_entropy_search(m, v, samples)
        return O.gibbon_quality_term(m, v, samples, st.noise) + O.gibbon_repulsion_term(st, x, pending, True)

    val, grad = O.entropy_value_and_grad(st, acq, samples, Xq, twin, 1.0 / 9.0)
    np.testing.assert_allclose(val, f(Xq), rtol=1e-8, atol=1e-12)
    assert np.abs(val).max() > 1e-3
    h = 1e-6
    num = np.stack([(f(Xq + h * e) - f(Xq - h * e)) / (2 * h) for e in np.eye(d)], axis=1)
    np.testing.assert_allclose(grad, num, rtol=1e-5, atol=1e-6 * np.abs(num).max())
