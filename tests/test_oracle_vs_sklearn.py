"""CPU: the oracle's exact-GPR posterior and marginal likelihood against an independent implementation --
scikit-learn's GaussianProcessRegressor (same model: constant-variance x stationary ARD kernel, Gaussian noise
on the diagonal, no jitter, no hyper-parameter optimisation).  A third leg of the oracle's pinning next to the
50-digit mpmath vectors (tests/golden) and the reference's own identities: GPflow cannot be installed here, but
the posterior of an exact GP is implementation-independent to rounding."""
import warnings

import numpy as np
import pytest

from oracle import gp_oracle as O

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern  # noqa: E402

KERNELS = {"rbf": lambda ls: RBF(length_scale=ls), "matern12": lambda ls: Matern(length_scale=ls, nu=0.5),
           "matern32": lambda ls: Matern(length_scale=ls, nu=1.5), "matern52": lambda ls: Matern(length_scale=ls, nu=2.5)}


@pytest.mark.parametrize("kind", sorted(KERNELS))
@pytest.mark.parametrize("d,N,noise", [(1, 7, 1e-2), (3, 40, 1e-3), (6, 120, 1e-1), (8, 200, 1e-2)])
def test_posterior_and_marginal_likelihood_match_scikit_learn(kind, d, N, noise):
    rng = np.random.default_rng(N * 10 + d)
    X = rng.uniform(size=(N, d))
    Y = np.sin(3.0 * X.sum(axis=1)) + 0.3 * rng.standard_normal(N)
    variance, c = 1.7, 0.25
    ls = 0.2 * np.sqrt(d) * (1.0 + 0.3 * np.arange(d))  # ARD
    st = O.gpr_update(kind, variance, ls, noise, c, X, Y)
    gpr = sk.GaussianProcessRegressor(kernel=ConstantKernel(variance, "fixed") * KERNELS[kind](ls), alpha=noise,
                                      optimizer=None, normalize_y=False)
    gpr.fit(X, Y - c)
    Xq = np.concatenate([rng.uniform(size=(50, d)), X[:3], 3.0 + rng.uniform(size=(2, d))])  # incl. training + far points
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # sklearn warns when a variance rounds below zero
        sm, sstd = gpr.predict(Xq, return_std=True)
        _, scov = gpr.predict(Xq[:20], return_cov=True)
    mean, var = O.predict(st, Xq, clip=False)
    floor = 64 * np.finfo(float).eps * variance * (1 + N * variance / noise)
    # the reference (like GPflow) forms r^2 = |a|^2 + |b|^2 - 2 a.b, sklearn takes differences first: for the
    # non-smooth Matern-1/2 that rounding difference in r reaches the values at the 1e-7 level
    rtol = 1e-6 if kind == "matern12" else 1e-8
    np.testing.assert_allclose(mean, sm + c, rtol=rtol, atol=floor * 10 + (1e-7 if kind == "matern12" else 0.0))
    np.testing.assert_allclose(np.maximum(var, 0.0), sstd ** 2, rtol=max(rtol, 1e-7), atol=floor * 10 + (1e-7 if kind == "matern12" else 0.0))
    jm, jc = O.predict_joint(st, Xq[:20])
    np.testing.assert_allclose(jc - np.diag(np.diag(jc)), scov - np.diag(np.diag(scov)), rtol=max(rtol, 1e-7),
                               atol=floor * 10 + (1e-7 if kind == "matern12" else 0.0))
    # -log p(y): the likelihood part of the training loss (models.py:256-292)
    nlml, _ = O.nlml_and_grad(st)
    np.testing.assert_allclose(-nlml, gpr.log_marginal_likelihood_value_, rtol=1e-9, atol=1e-8 if kind != "matern12" else 1e-5)
    # its gradient w.r.t. the log lengthscales, from sklearn's own analytic gradient at the same point
    free = sk.GaussianProcessRegressor(kernel=ConstantKernel(variance, "fixed") * KERNELS[kind](ls), alpha=noise,
                                       optimizer=None).fit(X, Y - c)
    _, sgrad = free.log_marginal_likelihood(free.kernel_.theta, eval_gradient=True)  # d/d log(ls)
    _, g = O.nlml_and_grad(st)
    np.testing.assert_allclose(-g[:d] * ls, sgrad, rtol=1e-6 if kind != "matern12" else 1e-4,
                               atol=1e-7 * max(1.0, np.abs(sgrad).max()))
