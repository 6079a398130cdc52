"""CPU: the oracle's analytic acquisition gradient vs central finite differences of its own values."""
import numpy as np
import pytest

from oracle import gp_oracle as O


@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("acq", ["ei", "pi", "nlcb"])
def test_gradient_matches_finite_differences(kind, acq):
    rng = np.random.default_rng(0)
    X, Y = O.synthetic_problem(O.hartmann_6, 6, 40)
    ls = O.default_lengthscales(6) * np.linspace(0.8, 1.4, 6)
    st = O.gpr_update(kind, 1.3, ls, 1e-2, float(Y.mean()), X, Y)
    Xq = rng.uniform(size=(6, 6))
    par = O.eta_min_mean(st) if acq != "nlcb" else 1.96
    tails = {"ei": O.expected_improvement, "pi": O.probability_of_improvement,
             "nlcb": O.negative_lower_confidence_bound}

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
