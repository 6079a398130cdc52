"""CPU oracle for the GP-posterior + acquisition hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy fp64 *restatement* of the algorithm that
secondmind-labs/trieste (reference @ 2024-10-16) runs for the path

    GaussianProcessRegression.update / predict / predict_joint
      -> expected_improvement / batch_monte_carlo_expected_improvement
      -> decoupled Thompson trajectories
      -> candidate sweep + arg-max.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product (``trieste_amd``) never does and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * The arithmetic of this path lives in third-party packages that are NOT vendored in
    /root/reference and NOT installable here: gpflow==2.9.2, gpflux==0.4.4,
    tensorflow==2.16.1, tensorflow-probability==0.24.0
    (reference tests/latest/constraints.txt:27,28,70,72).  Their published formulas are
    restated below, anchored on the reference's own call sites (cited per function).
  * The reference holds NO numeric golden vectors for this path (its tests are
    self-consistency identities and Monte-Carlo bounds).  The oracle is therefore pinned
    against (i) 50-digit mpmath golden vectors committed under tests/golden/ (generated
    by oracle/make_goldens.py) and (ii) every identity the reference's tests assert
    (tests/test_host_logic.py, tests/test_oracle_gradient.py) and (iii) scikit-learn's exact GPR on the same
    model (tests/test_oracle_vs_sklearn.py).  Bit-level parity with GPflow itself is UNPINNED
    and cannot be established in this container.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
from scipy.linalg import cholesky as _cholesky
from scipy.linalg import solve_triangular as _solve_triangular
from scipy.special import ndtr as _ndtr

JITTER = 1e-6  # reference trieste/utils/misc.py:180-183 (DEFAULTS.JITTER)
VAR_FLOOR = 1e-12  # reference trieste/models/gpflow/interface.py:123

KERNEL_KINDS = ("rbf", "matern12", "matern32", "matern52")


# --------------------------------------------------------------------------------------
# A.1 kernels (gpflow.kernels.Stationary / SquaredExponential / Matern{12,32,52}; chosen by
# reference trieste/models/gpflow/builders.py:383-408).  Call sites: models.py:212-215,
# sampler.py:677,690,843.
# --------------------------------------------------------------------------------------
def scaled_square_dist(X: np.ndarray, X2: np.ndarray, lengthscales: np.ndarray) -> np.ndarray:
    """gpflow ``Stationary.scaled_squared_euclid_dist``: inputs are divided by the
    lengthscales and r^2 = |a|^2 + |b|^2 - 2 a.b (gpflow.utilities.ops.square_distance)."""
    A = X / lengthscales
    B = X2 / lengthscales
    As = np.sum(A * A, axis=-1)[..., :, None]
    Bs = np.sum(B * B, axis=-1)[..., None, :]
    return As + Bs - 2.0 * np.matmul(A, np.swapaxes(B, -1, -2))


def kernel_from_r2(kind: str, variance: float, r2: np.ndarray) -> np.ndarray:
    """K(r^2).  SquaredExponential: variance*exp(-r2/2).  Matern: r = sqrt(max(r2, 1e-36))
    (gpflow ``IsotropicStationary.K_r2`` -> ``K_r``)."""
    if kind == "rbf":
        return variance * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, 1e-36))
    if kind == "matern12":
        return variance * np.exp(-r)
    if kind == "matern32":
        s3 = math.sqrt(3.0)
        return variance * (1.0 + s3 * r) * np.exp(-s3 * r)
    if kind == "matern52":
        s5 = math.sqrt(5.0)
        return variance * (1.0 + s5 * r + 5.0 / 3.0 * np.square(r)) * np.exp(-s5 * r)
    raise ValueError(f"unknown kernel kind {kind!r}")


def kernel_matrix(kind, variance, lengthscales, X, X2=None) -> np.ndarray:
    X = np.asarray(X, dtype=np.float64)
    X2 = X if X2 is None else np.asarray(X2, dtype=np.float64)
    ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (X.shape[-1],))
    return kernel_from_r2(kind, variance, scaled_square_dist(X, X2, ls))


# --------------------------------------------------------------------------------------
# A.2 update  (reference models/gpflow/models.py:171-186 -> interface.py:108-112 ->
# gpflow GPRPosterior._precompute: Kmm + sigma^2 I, cholesky (no jitter), err = Y - m(X))
# --------------------------------------------------------------------------------------
@dataclass
class GPRState:
    kind: str
    variance: float
    lengthscales: np.ndarray  # [d]
    noise: float
    mean_const: float
    X: np.ndarray  # [N, d]
    Y: np.ndarray  # [N]
    L: np.ndarray  # [N, N] lower Cholesky factor of K + noise*I
    err: np.ndarray  # [N]  Y - mean_const

    @property
    def N(self) -> int:
        return self.X.shape[0]

    @property
    def d(self) -> int:
        return self.X.shape[1]


def gpr_update(kind, variance, lengthscales, noise, mean_const, X, Y) -> GPRState:
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1)
    if X.ndim != 2 or X.shape[0] != Y.shape[0]:
        raise ValueError("X must be [N, d] and Y [N] / [N, 1]")
    ls = np.array(np.broadcast_to(np.asarray(lengthscales, dtype=np.float64), (X.shape[1],)))
    K = kernel_matrix(kind, variance, ls, X)
    K[np.diag_indices_from(K)] += noise
    L = _cholesky(K, lower=True)  # raises LinAlgError if not PD (TF: InvalidArgumentError)
    return GPRState(kind, float(variance), ls, float(noise), float(mean_const), X, Y, L,
                    Y - mean_const)


# --------------------------------------------------------------------------------------
# A.3 predict  (reference interface.py:119-124 + gpflow base_conditional_with_lm:
#   A = L^-1 Kmn;  fvar = Knn - sum(A^2);  A <- L^-T A;  fmean = A^T err (+ mean fn);
#   var clipped to [1e-12, max])
# --------------------------------------------------------------------------------------
def predict(state: GPRState, Xq: np.ndarray, clip: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    Xq = np.asarray(Xq, dtype=np.float64)
    lead = Xq.shape[:-1]
    Xf = Xq.reshape(-1, Xq.shape[-1])
    if Xf.shape[0] == 0:
        return np.zeros(lead), np.zeros(lead)
    Kmn = kernel_matrix(state.kind, state.variance, state.lengthscales, state.X, Xf)
    A = _solve_triangular(state.L, Kmn, lower=True)
    var = state.variance - np.sum(A * A, axis=0)
    A2 = _solve_triangular(state.L.T, A, lower=False)
    mean = A2.T @ state.err + state.mean_const
    if clip:
        var = np.clip(var, VAR_FLOOR, np.finfo(np.float64).max)
    return mean.reshape(lead), var.reshape(lead)


def predict_y(state: GPRState, Xq):
    """reference models.py:167-169: Gaussian likelihood adds the noise variance."""
    m, v = predict(state, Xq)
    return m, v + state.noise


def predict_joint(state: GPRState, Xq: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """reference interface.py:126-133.  Xq [..., q, d] -> mean [..., q], cov [..., q, q]
    (the reference's trailing/latent singleton axes are dropped); diag clipped at 1e-12."""
    Xq = np.asarray(Xq, dtype=np.float64)
    lead, q, d = Xq.shape[:-2], Xq.shape[-2], Xq.shape[-1]
    Xg = Xq.reshape(-1, q, d)
    G = Xg.shape[0]
    means = np.empty((G, q))
    covs = np.empty((G, q, q))
    for g in range(G):
        Kmn = kernel_matrix(state.kind, state.variance, state.lengthscales, state.X, Xg[g])
        Knn = kernel_matrix(state.kind, state.variance, state.lengthscales, Xg[g])
        A = _solve_triangular(state.L, Kmn, lower=True)
        cov = Knn - A.T @ A
        A2 = _solve_triangular(state.L.T, A, lower=False)
        means[g] = A2.T @ state.err + state.mean_const
        dg = np.clip(np.diag(cov), VAR_FLOOR, np.finfo(np.float64).max)
        cov[np.diag_indices(q)] = dg
        covs[g] = cov
    return means.reshape(lead + (q,)), covs.reshape(lead + (q, q))


def covariance_between_points(state: GPRState, X1: np.ndarray, X2: np.ndarray) -> np.ndarray:
    """reference models/gpflow/models.py:188-254:  K12 - (L^-1 Kx1)^T (L^-1 Kx2)."""
    Kx1 = kernel_matrix(state.kind, state.variance, state.lengthscales, state.X, X1)
    Kx2 = kernel_matrix(state.kind, state.variance, state.lengthscales, state.X, X2)
    K12 = kernel_matrix(state.kind, state.variance, state.lengthscales, X1, X2)
    A1 = _solve_triangular(state.L, Kx1, lower=True)
    A2 = _solve_triangular(state.L, Kx2, lower=True)
    return K12 - A1.T @ A2


# --------------------------------------------------------------------------------------
# A.4 eta and the acquisition tails (reference acquisition/function/function.py)
# --------------------------------------------------------------------------------------
def eta_min_mean(state: GPRState, Xtrain: Optional[np.ndarray] = None) -> float:
    """function.py:145-149: eta = min over the dataset's query points of the posterior MEAN."""
    m, _ = predict(state, state.X if Xtrain is None else Xtrain)
    return float(np.min(m))


def normal_cdf(z):
    return _ndtr(z)


def normal_pdf(z):
    return np.exp(-0.5 * np.square(z)) / math.sqrt(2.0 * math.pi)


def expected_improvement(mean, var, eta):
    """function.py:220-223: normal = Normal(mean, sqrt(var));
    (eta-mean)*normal.cdf(eta) + var*normal.prob(eta)."""
    mean = np.asarray(mean, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    sd = np.sqrt(var)
    z = (eta - mean) / sd
    return (eta - mean) * normal_cdf(z) + var * (normal_pdf(z) / sd)


def probability_of_improvement(mean, var, threshold):
    """function.py:509-510: Normal(mean, sqrt(var)).cdf(threshold)."""
    return normal_cdf((threshold - np.asarray(mean)) / np.sqrt(var))


def negative_lower_confidence_bound(mean, var, beta=1.96):
    """function.py:389-418: -(mean - beta*sqrt(var))."""
    return -(np.asarray(mean) - beta * np.sqrt(var))


def augmented_expected_improvement(mean, var, eta, noise):
    """function.py:312-325: EI * (1 - sqrt(noise) / sqrt(noise + var))."""
    var = np.asarray(var, dtype=np.float64)
    return expected_improvement(mean, var, eta) * (1.0 - math.sqrt(noise) / np.sqrt(noise + var))


def ei_values(state: GPRState, Xq: np.ndarray, eta: float) -> np.ndarray:
    m, v = predict(state, Xq)
    return expected_improvement(m, v, eta)


def _kernel_dr2(kind, variance, r2):
    """d k / d (r^2) of the stationary kernels (analytic; the reference gets it by TF autodiff)."""
    if kind == "rbf":
        return -0.5 * variance * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, 1e-36))
    if kind == "matern12":
        return -0.5 * variance * np.exp(-r) / r
    if kind == "matern32":
        return -1.5 * variance * np.exp(-math.sqrt(3.0) * r)
    s = math.sqrt(5.0) * r
    return -(5.0 / 6.0) * variance * (1.0 + s) * np.exp(-s)


def acq_value_and_grad(state: GPRState, acq: str, param: float, Xq: np.ndarray):
    """Acquisition value [P] and gradient [P, d] w.r.t. the query points -- the quantity
    tfp.math.value_and_gradient produces in the reference's L-BFGS-B refinement
    (acquisition/optimizer.py:628-629) -- in analytic form: d mean = sum_k alpha_k dk_k,
    d var = -2 (K^-1 k*)^T dk*, chain rule through EI / PI / -LCB; the variance clip has zero
    gradient (tf.clip_by_value)."""
    Xq = np.asarray(Xq, dtype=np.float64)
    ls = state.lengthscales
    A = Xq / ls
    B = state.X / ls
    diff = A[:, None, :] - B[None, :, :]                      # [P, N, d]
    r2 = np.sum(diff * diff, axis=-1)                          # difference form (exact zero at coincidence)
    Kq = kernel_from_r2(state.kind, state.variance, r2)        # [P, N]
    alpha = _solve_triangular(state.L.T, _solve_triangular(state.L, state.err, lower=True), lower=False)
    Z = _solve_triangular(state.L.T, _solve_triangular(state.L, Kq.T, lower=True), lower=False).T  # [P, N]
    mu = Kq @ alpha + state.mean_const
    var_raw = state.variance - np.sum(Kq * Z, axis=1)
    clipped = ~(var_raw > VAR_FLOOR)
    var = np.where(clipped, VAR_FLOOR, var_raw)
    dk = (2.0 * _kernel_dr2(state.kind, state.variance, r2))[:, :, None] * diff / ls  # [P, N, d]
    dmu = np.einsum("pnd,n->pd", dk, alpha)
    dvar = np.where(clipped[:, None], 0.0, -2.0 * np.einsum("pnd,pn->pd", dk, Z))
    sd = np.sqrt(var)
    if acq == "ei":
        z = (param - mu) / sd
        val = (param - mu) * normal_cdf(z) + sd * normal_pdf(z)
        g = -normal_cdf(z)[:, None] * dmu + (normal_pdf(z) / (2.0 * sd))[:, None] * dvar
    elif acq == "pi":
        z = (param - mu) / sd
        val = normal_cdf(z)
        g = (-normal_pdf(z) / sd)[:, None] * dmu + (-normal_pdf(z) * z / (2.0 * var))[:, None] * dvar
    elif acq == "nlcb":
        val = -(mu - param * sd)
        g = -dmu + (param / (2.0 * sd))[:, None] * dvar
    elif acq == "aei":
        z = (param - mu) / sd
        ei = (param - mu) * normal_cdf(z) + sd * normal_pdf(z)
        sn, st = math.sqrt(state.noise), np.sqrt(state.noise + var)
        aug = 1.0 - sn / st
        daug = 0.5 * sn / (st * (state.noise + var))
        val = ei * aug
        g = (-normal_cdf(z) * aug)[:, None] * dmu + (normal_pdf(z) / (2.0 * sd) * aug + ei * daug)[:, None] * dvar
    else:
        raise KeyError(acq)
    return val, g


def nlml_and_grad(state: GPRState):
    """Negative log marginal likelihood of (hyper-parameters, data) and its gradient w.r.t.
    (lengthscales [d], variance, noise, mean) -- the likelihood part of gpflow GPR.training_loss that
    GaussianProcessRegression.optimize_encoded minimises (reference models/gpflow/models.py:256-292).
    nlml = 1/2 err^T K^-1 err + sum log L_ii + N/2 log 2 pi;  d/d theta = 1/2 tr((K^-1 - a a^T) dK/d theta)."""
    N, d = state.N, state.d
    ls = state.lengthscales
    alpha = _solve_triangular(state.L.T, _solve_triangular(state.L, state.err, lower=True), lower=False)
    Linv = _solve_triangular(state.L, np.eye(N), lower=True)
    G = Linv.T @ Linv - np.outer(alpha, alpha)
    value = 0.5 * state.err @ alpha + np.sum(np.log(np.diag(state.L))) + 0.5 * N * math.log(2.0 * math.pi)
    A = state.X / ls
    diff2 = (A[:, None, :] - A[None, :, :]) ** 2          # [N, N, d] scaled squared differences
    r2 = diff2.sum(-1)
    Kf = kernel_from_r2(state.kind, state.variance, r2)
    f1 = _kernel_dr2(state.kind, state.variance, r2)
    grad = np.empty(d + 3)
    for c in range(d):
        grad[c] = 0.5 * np.sum(G * f1 * (-2.0 * diff2[:, :, c] / ls[c]))
    grad[d] = 0.5 * np.sum(G * Kf) / state.variance
    grad[d + 1] = 0.5 * np.trace(G)
    grad[d + 2] = -np.sum(alpha)
    return float(value), grad


# --------------------------------------------------------------------------------------
# A.5 batch Monte-Carlo EI (sampler.py:208-287 + function.py:1181-1186); eps passed in.
# --------------------------------------------------------------------------------------
def batch_reparam_samples(state: GPRState, Xq: np.ndarray, eps: np.ndarray,
                          jitter: float = JITTER) -> np.ndarray:
    """Xq [G, q, d], eps [q, S] -> samples [G, S, q]:  mean + (chol(cov + jitter I) @ eps)^T."""
    mean, cov = predict_joint(state, Xq)
    G, q = mean.shape
    out = np.empty((G, eps.shape[1], q))
    for g in range(G):
        Lq = _cholesky(cov[g] + jitter * np.eye(q), lower=True)
        out[g] = mean[g][None, :] + (Lq @ eps).T
    return out


def batch_mc_ei(state: GPRState, Xq: np.ndarray, eps: np.ndarray, eta: float,
                jitter: float = JITTER) -> np.ndarray:
    """qEI [G] = mean_S max(eta - min_q samples, 0)."""
    s = batch_reparam_samples(state, Xq, eps, jitter)  # [G, S, q]
    return np.mean(np.maximum(eta - np.min(s, axis=-1), 0.0), axis=-1)


def batch_mc_ei_value_and_grad(state: GPRState, Xq: np.ndarray, eps: np.ndarray, eta: float, jitter: float = JITTER):
    """qEI [G] and its gradient [G, q, d] w.r.t. the batch points -- what tfp.math.value_and_gradient
    (reference acquisition/optimizer.py:628-629) returns when batchify_joint (optimizer.py:897-934) hands
    BatchMonteCarloExpectedImprovement (function.py:1150-1186) to the continuous optimizer: the derivative through predict_joint
    (interface.py:126-133, a clipped diagonal entry has zero gradient: tf.clip_by_value), tf.linalg.cholesky and the
    reparametrised samples (sampler.py:276-287).  FORWARD mode, one tangent per coordinate (i, c): d mean, d cov from the
    kernel derivatives, d L = L Phi(L^-1 d cov L^-T) (Phi: lower triangle, diagonal halved), d samples = d mean + d L eps,
    d value = mean_s 1[improvement > 0] (-d sample at the arg-min).  Plain loops: small cases only."""
    Xq = np.asarray(Xq, dtype=np.float64)
    G, q, d = Xq.shape
    ls = state.lengthscales
    Bx = state.X / ls
    alpha = _solve_triangular(state.L.T, _solve_triangular(state.L, state.err, lower=True), lower=False)
    S = eps.shape[1]
    vals = np.empty(G)
    grads = np.zeros((G, q, d))
    for g in range(G):
        A = Xq[g] / ls
        diff = A[:, None, :] - Bx[None, :, :]                      # [q, N, d]
        r2 = np.sum(diff * diff, axis=-1)
        Kq = kernel_from_r2(state.kind, state.variance, r2)        # [q, N]
        Cq = _solve_triangular(state.L, Kq.T, lower=True)          # [N, q]  c_i = L^-1 k*_i
        dff = A[:, None, :] - A[None, :, :]                        # [q, q, d]
        r2q = np.sum(dff * dff, axis=-1)
        cov = kernel_from_r2(state.kind, state.variance, r2q) - Cq.T @ Cq
        raw_diag = np.diag(cov).copy()
        clipped = ~(raw_diag > VAR_FLOOR)
        cov[np.diag_indices(q)] = np.where(clipped, VAR_FLOOR, raw_diag)
        mean = Kq @ alpha + state.mean_const
        Lq = _cholesky(cov + jitter * np.eye(q), lower=True)
        smp = mean[None, :] + (Lq @ eps).T                          # [S, q]
        jmin = np.argmin(smp, axis=1)
        imp = eta - smp[np.arange(S), jmin]
        active = imp > 0.0
        vals[g] = np.mean(np.where(active, imp, 0.0))
        dk = (2.0 * _kernel_dr2(state.kind, state.variance, r2))[:, :, None] * diff / ls       # [q, N, d]  d k*_i / d x_i
        dkq = (2.0 * _kernel_dr2(state.kind, state.variance, r2q))[:, :, None] * dff / ls      # [q, q, d]  d k(x_i, x_j) / d x_i
        for i in range(q):
            for c in range(d):
                dmean = np.zeros(q)
                dmean[i] = dk[i, :, c] @ alpha
                dci = _solve_triangular(state.L, dk[i, :, c], lower=True)   # d c_i
                dcov = np.zeros((q, q))
                row = -(dci @ Cq)                                           # -d c_i^T c_j for every j
                for j in range(q):
                    if j != i:
                        row[j] += dkq[i, j, c]
                dcov[i, :] += row
                dcov[:, i] += row                                           # (entry (i, i) twice: -2 d c_i^T c_i)
                if clipped[i]:
                    dcov[i, i] = 0.0
                M = _solve_triangular(Lq, _solve_triangular(Lq, dcov, lower=True).T, lower=True).T   # L^-1 dcov L^-T
                Phi = np.tril(M)
                Phi[np.diag_indices(q)] *= 0.5
                dL = Lq @ Phi
                dsmp = dmean[None, :] + (dL @ eps).T                        # [S, q]
                grads[g, i, c] = np.mean(np.where(active, -dsmp[np.arange(S), jmin], 0.0))
    return vals, grads


def joint_samples(state: GPRState, Xq: np.ndarray, eps: np.ndarray, jitter: float = JITTER) -> np.ndarray:
    """GPflowPredictor.sample_encoded (interface.py:135-137) -> gpflow predict_f_samples: mean, cov =
    predict_f(full_cov=True) (NO clipping), samples = mean + chol(cov + jitter I) eps (sample_mvn).
    Xq [n, d], eps [n, S] -> [S, n]."""
    Xq = np.asarray(Xq, dtype=np.float64)
    m, _ = predict(state, Xq)
    cov = covariance_between_points(state, Xq, Xq)
    cov = 0.5 * (cov + cov.T)
    Lc = _cholesky(cov + jitter * np.eye(Xq.shape[0]), lower=True)
    return (m[:, None] + Lc @ np.asarray(eps, dtype=np.float64)).T


def independent_reparam_samples(state: GPRState, Xq: np.ndarray, eps: np.ndarray,
                                jitter: float = JITTER) -> np.ndarray:
    """IndependentReparametrizationSampler.sample (sampler.py:117-164): Xq [M, d], eps [S] ->
    samples [M, S] = mean + sqrt(var + jitter) * eps (marginal posteriors, no cross-covariance)."""
    m, v = predict(state, Xq)
    return m[:, None] + np.sqrt(v + jitter)[:, None] * np.asarray(eps, dtype=np.float64)[None, :]


def conditional_predict_joint(state: GPRState, Xq: np.ndarray, X_add: np.ndarray, Y_add: np.ndarray):
    """models.py:418-484: joint posterior at Xq [M, d] conditioned also on additional (noisy)
    observations (X_add [n, d], Y_add [n]):  from the joint predict at [X_add; Xq],
    L_add = chol(cov_add + noise I), A = L_add^-1 cov_cross, cov' = cov_qp - A^T A,
    mean' = mean_qp + A^T L_add^-1 (Y_add - mean_add)."""
    X_add = np.asarray(X_add, dtype=np.float64)
    n = X_add.shape[0]
    mean, cov = predict_joint(state, np.concatenate([X_add, np.asarray(Xq, dtype=np.float64)], axis=0))
    L_add = _cholesky(cov[:n, :n] + state.noise * np.eye(n), lower=True)
    A = _solve_triangular(L_add, cov[:n, n:], lower=True)
    AM = _solve_triangular(L_add, np.asarray(Y_add, dtype=np.float64).reshape(n) - mean[:n], lower=True)
    return mean[n:] + A.T @ AM, cov[n:, n:] - A.T @ A


def conditional_predict_f(state: GPRState, Xq: np.ndarray, X_add: np.ndarray, Y_add: np.ndarray):
    """models.py:355-416: marginal version; uses predict (CLIPPED variance) at Xq, predict_joint
    at X_add and covariance_between_points(X_add, Xq)."""
    X_add = np.asarray(X_add, dtype=np.float64)
    n = X_add.shape[0]
    mean_add, cov_add = predict_joint(state, X_add)
    mean_qp, var_qp = predict(state, Xq)
    cross = covariance_between_points(state, X_add, np.asarray(Xq, dtype=np.float64))  # [n, M]
    L_add = _cholesky(cov_add + state.noise * np.eye(n), lower=True)
    A = _solve_triangular(L_add, cross, lower=True)
    AM = _solve_triangular(L_add, np.asarray(Y_add, dtype=np.float64).reshape(n) - mean_add, lower=True)
    return mean_qp + A.T @ AM, var_qp - np.sum(A * A, axis=0)


# --------------------------------------------------------------------------------------
# Greedy batches on the same posterior (acquisition/function/greedy_batch.py)
# --------------------------------------------------------------------------------------
def lipschitz_estimate(state: GPRState, points: np.ndarray) -> Tuple[float, float]:
    """LocalPenalization._get_lipschitz_estimate (greedy_batch.py:206-217): (max_i |d mean / d x_i|_2,
    min_i mean(x_i)) over the sampled points; the gradient (autodiff there) in analytic form."""
    val, g = acq_value_and_grad(state, "nlcb", 0.0, points)  # -(mean - 0 * sd) = -mean
    return float(np.max(np.linalg.norm(g, axis=1))), float(np.min(-val))


def local_penalizer_parameters(state: GPRState, pending: np.ndarray, lipschitz: float, eta: float):
    """local_penalizer.__init__ (greedy_batch.py:287-300): radius = (mean(pending) - eta) / L,
    scale = sqrt(var(pending)) / L (predict's clipped variance)."""
    mean, var = predict(state, pending)
    return (mean - eta) / lipschitz, np.sqrt(var) / lipschitz


def _pairwise_distances(x: np.ndarray, pending: np.ndarray) -> np.ndarray:
    x, pending = np.asarray(x, dtype=np.float64), np.asarray(pending, dtype=np.float64)
    return np.linalg.norm(x[:, None, :] - pending[None, :, :], axis=-1)  # [M, P]


def soft_local_penalizer(x, pending, radius, scale) -> np.ndarray:
    """soft_local_penalizer.__call__ (greedy_batch.py:341-354): prod_p Phi((|x - x_p| - r_p) / s_p); x [M, d]."""
    z = (_pairwise_distances(x, pending) - np.asarray(radius)[None, :]) / np.asarray(scale)[None, :]
    return np.prod(normal_cdf(z), axis=-1)


def hard_local_penalizer(x, pending, radius, scale) -> np.ndarray:
    """hard_local_penalizer.__call__ (greedy_batch.py:376-389): prod_p ((|x - x_p| / (r_p + s_p))^p + 1)^(1/p),
    p = -5."""
    p = -5.0
    with np.errstate(divide="ignore"):
        pen = ((_pairwise_distances(x, pending) / (np.asarray(radius) + np.asarray(scale))[None, :]) ** p + 1.0) ** (1.0 / p)
    return np.prod(pen, axis=-1)


PENALIZERS = {"soft": soft_local_penalizer, "hard": hard_local_penalizer}


def penalized_acquisition(base_values: np.ndarray, penalization: np.ndarray) -> np.ndarray:
    """PenalizedAcquisition.__call__ (greedy_batch.py:265-269): exp(log a + log phi)."""
    with np.errstate(divide="ignore"):
        return np.exp(np.log(base_values) + np.log(penalization))


def penalized_value_and_grad(state: GPRState, acq: str, param: float, kind: str, pending, radius, scale, Xq):
    """Value [P'] and gradient [P', d] of a(x) prod_p phi_p(x) -- what autodiff through PenalizedAcquisition gives
    the L-BFGS-B refinement -- analytically: phi a' + a sum_p phi_p' prod_{q != p} phi_q, with
    d Phi(z_p) / dx = pdf(z_p) / s_p * (x - x_p) / dist_p (soft) and
    d ((u^-5 + 1)^(-1/5)) / dx = u^-6 (u^-5 + 1)^(-6/5) / (r_p + s_p) * (x - x_p) / dist_p (hard, u = dist_p / (r_p + s_p))."""
    Xq = np.asarray(Xq, dtype=np.float64)
    pending = np.asarray(pending, dtype=np.float64)
    radius, scale = np.asarray(radius, dtype=np.float64), np.asarray(scale, dtype=np.float64)
    a, da = acq_value_and_grad(state, acq, param, Xq)
    diff = Xq[:, None, :] - pending[None, :, :]             # [P', P, d]
    dist = np.linalg.norm(diff, axis=-1)                      # [P', P]
    if kind == "soft":
        z = (dist - radius) / scale
        phi = normal_cdf(z)
        slope = normal_pdf(z) / scale
    else:
        u = dist / (radius + scale)
        with np.errstate(divide="ignore", invalid="ignore"):
            phi = (u ** -5.0 + 1.0) ** -0.2
            slope = u ** -6.0 * (u ** -5.0 + 1.0) ** -1.2 / (radius + scale)
    P = pending.shape[0]
    others = np.stack([np.prod(np.delete(phi, p, axis=1), axis=1) for p in range(P)], axis=1)  # [P', P]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dist > 0.0, slope * others / dist, 0.0)
    dphi = np.einsum("mp,mpd->md", w, diff)
    prod = np.prod(phi, axis=1)
    return a * prod, prod[:, None] * da + a[:, None] * dphi


def fantasized_state(state: GPRState, X_add: np.ndarray, Y_add: np.ndarray) -> GPRState:
    """The posterior _fantasized_model represents (greedy_batch.py:630-773: the base model's
    conditional_predict_* with the fantasised data): for an exact GPR with Gaussian noise this IS the GPR on
    (data + fantasised data) with the same hyper-parameters."""
    X = np.concatenate([state.X, np.asarray(X_add, dtype=np.float64)], axis=0)
    Y = np.concatenate([state.Y, np.asarray(Y_add, dtype=np.float64).reshape(-1)])
    return gpr_update(state.kind, state.variance, state.lengthscales, state.noise, state.mean_const, X, Y)


# --------------------------------------------------------------------------------------
# Entropy search on the same posterior (acquisition/function/entropy.py, acquisition/sampler.py:126-213)
# --------------------------------------------------------------------------------------
CLAMP_LB = 1e-8  # entropy.py:47
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def log_normal_cdf(x):
    """tfp.distributions.Normal(0, 1).log_cdf = special_math.log_ndtr for float64 (tensorflow-probability 0.24,
    special_math.py): x > 8: -ndtr(-x);  -20 <= x <= 8: log(ndtr(x));  x < -20: asymptotic series of order 3,
    -x^2/2 - log(-x) - log(2 pi)/2 + log(1 - 1/x^2 + 3/x^4 - 15/x^6)."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mid = np.log(_ndtr(np.clip(x, -20.0, 8.0)))
        hi = -_ndtr(-np.maximum(x, 8.0))
        xl = np.minimum(x, -20.0)
        x2 = xl * xl
        lo = -0.5 * x2 - np.log(-xl) - _HALF_LOG_2PI + np.log(1.0 - 1.0 / x2 + 3.0 / x2 ** 2 - 15.0 / x2 ** 3)
    return np.where(x > 8.0, hi, np.where(x >= -20.0, mid, lo))


def _gamma_ratio(mean, var, samples):
    fsd = np.maximum(np.sqrt(var), CLAMP_LB)                       # clip below (entropy.py:201-204)
    gamma = (np.asarray(samples, dtype=np.float64).reshape(1, -1) - np.asarray(mean)[:, None]) / fsd[:, None]
    log_minus_cdf = log_normal_cdf(-gamma)
    ratio = np.exp(-0.5 * gamma * gamma - _HALF_LOG_2PI - log_minus_cdf)   # exp(log_prob(gamma) - log_cdf(-gamma))
    return gamma, log_minus_cdf, ratio


def min_value_entropy_search(mean, var, samples) -> np.ndarray:
    """min_value_entropy_search.__call__ (entropy.py:195-214): mean [M], var [M] (predict's), samples [S]."""
    gamma, log_minus_cdf, ratio = _gamma_ratio(np.asarray(mean), np.asarray(var), samples)
    return np.mean(-gamma * ratio / 2.0 - log_minus_cdf, axis=1)


def gibbon_quality_term(mean, var, samples, noise) -> np.ndarray:
    """gibbon_quality_term.__call__ (entropy.py:479-500)."""
    var = np.asarray(var, dtype=np.float64)
    rho_squared = var / (var + noise)
    gamma, _, ratio = _gamma_ratio(np.asarray(mean), var, samples)
    inner_log = 1.0 + rho_squared[:, None] * ratio * (gamma - ratio)
    return -0.5 * np.mean(np.log(inner_log), axis=1)


def gibbon_repulsion_term(state: GPRState, x: np.ndarray, pending: np.ndarray, rescaled_repulsion: bool = True):
    """gibbon_repulsion_term.__call__ (entropy.py:580-619): yvar = predict var + noise; B = predict_joint cov of
    the pending points; L = chol(B + noise I); A = covariance_between_points(x, pending);
    V_det = yvar - |L^-1 A|^2; 1/2 (log V_det - log yvar), times (1/m)^2 if rescaled."""
    x, pending = np.asarray(x, dtype=np.float64), np.asarray(pending, dtype=np.float64)
    _, fvar = predict(state, x)
    yvar = fvar + state.noise
    _, B = predict_joint(state, pending)
    m = pending.shape[0]
    L = _cholesky(B + state.noise * np.eye(m), lower=True)
    A = covariance_between_points(state, x, pending)              # [M, m]
    L_inv_A = _solve_triangular(L, A.T, lower=True)               # [m, M]
    V_det = yvar - np.sum(L_inv_A * L_inv_A, axis=0)
    repulsion = 0.5 * (np.log(V_det) - np.log(yvar))
    return repulsion * ((1.0 / m) ** 2 if rescaled_repulsion else 1.0)


def entropy_value_and_grad(state: GPRState, acq: str, samples, Xq, twin: Optional[GPRState] = None,
                           weight: float = 0.0):
    """Value [P] and gradient [P, d] of the MES / GIBBON acquisition at Xq -- what autodiff through
    min_value_entropy_search / GibbonAcquisition hands L-BFGS-B -- analytically.  With u = gamma,
    r = pdf(u) / Phi(-u): dr/du = r (r - u);  MES: f = -u r / 2 - log Phi(-u), df/du = r/2 - u r (r - u) / 2;
    GIBBON: g = -1/2 log(1 + rho^2 h), h = r (u - r), dh/du = -r (u - r)^2 + r - r^2 (r - u);
    du/dmean = -1/sd, du/dvar = -u / (2 var), d rho^2 / dvar = noise / (var + noise)^2.  The repulsion term (twin =
    the state conditioned on the pending points) adds weight/2 (log(var_twin + noise) - log(var + noise))."""
    Xq = np.asarray(Xq, dtype=np.float64)

    def mean_var_grads(st):
        negm, dnegm = acq_value_and_grad(st, "nlcb", 0.0, Xq)       # -mean, -dmean
        v1, g1 = acq_value_and_grad(st, "nlcb", 1.0, Xq)            # -(mean - sd)
        sd = v1 - negm
        dsd = g1 - dnegm
        return -negm, sd * sd, -dnegm, 2.0 * sd[:, None] * dsd

    mu, var, dmu, dvar = mean_var_grads(state)
    sd_raw = np.sqrt(var)
    clamped = ~(sd_raw > CLAMP_LB)
    gamma, lmc, r = _gamma_ratio(mu, var, samples)
    if acq == "mes":
        f = -gamma * r / 2.0 - lmc
        fu = r / 2.0 - gamma * r * (r - gamma) / 2.0
        frho = np.zeros_like(f)
    else:
        rho2 = (var / (var + state.noise))[:, None]
        h = r * (gamma - r)
        inner = 1.0 + rho2 * h
        dh = -r * (gamma - r) ** 2 + r - r * r * (r - gamma)
        f = -0.5 * np.log(inner)
        fu = -0.5 * rho2 * dh / inner
        frho = -0.5 * h / inner
    sd = np.maximum(sd_raw, CLAMP_LB)
    val = f.mean(axis=1)
    dv_dmu = -fu.mean(axis=1) / sd
    dv_dvar = np.where(clamped, 0.0, -(fu * gamma).mean(axis=1) / (2.0 * var))
    if acq != "mes":
        dv_dvar = dv_dvar + frho.mean(axis=1) * state.noise / (var + state.noise) ** 2
    grad = dv_dmu[:, None] * dmu + dv_dvar[:, None] * dvar
    if twin is not None and acq != "mes":
        _, vt, _, dvt = mean_var_grads(twin)
        val = val + 0.5 * weight * (np.log(vt + state.noise) - np.log(var + state.noise))
        grad = grad + 0.5 * weight * (dvt / (vt + state.noise)[:, None] - dvar / (var + state.noise)[:, None])
    return val, grad


def gumbel_min_value_samples(mean, sd, uniform_samples) -> np.ndarray:
    """GumbelSampler.sample (acquisition/sampler.py:156-212) given the model's (predict_y) mean / sd at the grid
    and the uniform draws: fit a Gumbel to the quartiles of Pr(y* < y) = 1 - prod_i Phi(-(y - mean_i) / sd_i)
    (bisection on [min(mean - 5 sd), max(mean + 5 sd)]), then y = a + b log(-log(1 - u))."""
    from scipy.optimize import bisect

    mean, sd = np.asarray(mean, dtype=np.float64).reshape(-1), np.asarray(sd, dtype=np.float64).reshape(-1)

    def probf(y):
        return 1.0 - np.exp(np.sum(log_normal_cdf(-(y - mean) / sd)))

    left, right = float(np.min(mean - 5.0 * sd)), float(np.max(mean + 5.0 * sd))
    q1, q2 = (bisect(lambda y: probf(y) - val, left, right, maxiter=10000) for val in (0.25, 0.75))
    l1, l2 = math.log(math.log(4.0 / 3.0)), math.log(math.log(4.0))
    b = (q1 - q2) / (l1 - l2)
    a = (q2 * l1 - q1 * l2) / (l1 - l2)
    u = np.asarray(uniform_samples, dtype=np.float64).reshape(-1)
    return (np.log(-np.log(1.0 - u)) * b + a)[:, None]


# --------------------------------------------------------------------------------------
# A.6 decoupled trajectories (sampler.py:661-738, 801-806, 841-855, 901-936;
# gpflux RandomFourierFeaturesCosine, gpflux.math.compute_A_inv_b); draws passed in.
# --------------------------------------------------------------------------------------
def rff_features(state: GPRState, Xq: np.ndarray, W: np.ndarray, b: np.ndarray) -> np.ndarray:
    """phi(x) = sqrt(2 variance / F) * cos((x / ls) W^T + b), W [F, d], b [F] -> [M, F]."""
    F = W.shape[0]
    return math.sqrt(2.0 * state.variance / F) * np.cos((Xq / state.lengthscales) @ W.T + b)


def decoupled_weights(state: GPRState, W, b, w, xi) -> np.ndarray:
    """sampler.py:702-736 (exact-GP branch 679-690): u = (Y - c) + sqrt(noise)*xi;
    v = (K + noise I)^-1 (u - Phi_Z w).  w [F, B], xi [N, B] -> v [N, B]."""
    w = np.asarray(w, dtype=np.float64).reshape(W.shape[0], -1)
    xi = np.asarray(xi, dtype=np.float64).reshape(state.N, -1)
    u = state.err[:, None] + math.sqrt(state.noise) * xi
    phiZ = rff_features(state, state.X, W, b)
    diff = u - phiZ @ w
    # compute_A_inv_b does its own Cholesky of Kmm (= K + noise I, identical to state.L)
    t = _solve_triangular(state.L, diff, lower=True)
    return _solve_triangular(state.L.T, t, lower=False)


def rff_theta(state: GPRState, W, b, eps) -> np.ndarray:
    """RandomFourierFeatureTrajectorySampler (sampler.py:518-591): weights theta [F, B] of the scaled
    features = posterior mean + chol(posterior cov) eps, eps [F, B].
    F < N ("design space", 529-556): D = Phi^T Phi + noise I, D^-1 by cholesky_solve,
    mean = D^-1 Phi^T r, cov = noise D^-1.   N <= F ("gram space", 558-591): G = Phi Phi^T + noise I,
    A = L_G^-1 Phi, mean = A^T L_G^-1 r, cov = I - A^T A."""
    F = W.shape[0]
    eps = np.asarray(eps, dtype=np.float64).reshape(F, -1)
    phi = rff_features(state, state.X, W, b)  # [N, F]
    r = state.err
    if F < state.N:
        Ld = _cholesky(phi.T @ phi + state.noise * np.eye(F), lower=True)
        Dinv = _solve_triangular(Ld.T, _solve_triangular(Ld, np.eye(F), lower=True), lower=False)
        mean = Dinv @ (phi.T @ r)
        cov = state.noise * Dinv
    else:
        Lg = _cholesky(phi @ phi.T + state.noise * np.eye(state.N), lower=True)
        A = _solve_triangular(Lg, phi, lower=True)
        mean = A.T @ _solve_triangular(Lg, r, lower=True)
        cov = np.eye(F) - A.T @ A
    Lc = _cholesky(0.5 * (cov + cov.T), lower=True)
    return mean[:, None] + Lc @ eps


def rff_trajectory_eval(state: GPRState, W, b, theta, Xq: np.ndarray) -> np.ndarray:
    """f_b(x) = phi(x) . theta_b + c (feature_decomposition_trajectory.__call__, sampler.py:923-936, with
    the RFF-only feature functions); Xq [M, d] shared or [M, B, d] -> [M, B]."""
    theta = np.asarray(theta, dtype=np.float64).reshape(W.shape[0], -1)
    Xq = np.asarray(Xq, dtype=np.float64)
    if Xq.ndim == 2:
        return rff_features(state, Xq, W, b) @ theta + state.mean_const
    return np.stack([rff_features(state, Xq[:, bb, :], W, b) @ theta[:, bb] for bb in range(theta.shape[1])],
                    axis=1) + state.mean_const


def trajectory_eval(state: GPRState, W, b, w, v, Xq: np.ndarray) -> np.ndarray:
    """sampler.py:923-936: f(x)_b = phi(x).w_b + k(x, X).v_b + c.   Xq [M, d] (shared by all
    B trajectories) or [M, B, d] -> [M, B]."""
    w = np.asarray(w, dtype=np.float64).reshape(W.shape[0], -1)
    v = np.asarray(v, dtype=np.float64).reshape(state.N, -1)
    B = w.shape[1]
    Xq = np.asarray(Xq, dtype=np.float64)
    if Xq.ndim == 2:
        phi = rff_features(state, Xq, W, b)
        Kx = kernel_matrix(state.kind, state.variance, state.lengthscales, Xq, state.X)
        return phi @ w + Kx @ v + state.mean_const
    out = np.empty((Xq.shape[0], B))
    for bb in range(B):
        phi = rff_features(state, Xq[:, bb, :], W, b)
        Kx = kernel_matrix(state.kind, state.variance, state.lengthscales, Xq[:, bb, :], state.X)
        out[:, bb] = phi @ w[:, bb] + Kx @ v[:, bb] + state.mean_const
    return out


def trajectory_value_and_grad(state: GPRState, W, b, w, v, Xq: np.ndarray):
    """Value [P, B] and gradient [P, B, d] of the trajectories at per-trajectory points Xq [P, B, d]
    -- what tfp.math.value_and_gradient gives the L-BFGS-B refinement of
    Greedy/ParallelContinuousThompsonSampling (continuous_thompson_sampling.py:30-245 through
    acquisition/optimizer.py:628-629); analytic: d phi_f = -sqrt(2 var / F) sin(.) W_f / ls,
    d k(x, X_k) = 2 k'(r2) (x - X_k) / ls^2."""
    w = np.asarray(w, dtype=np.float64).reshape(W.shape[0], -1)
    v = np.asarray(v, dtype=np.float64).reshape(state.N, -1)
    Xq = np.asarray(Xq, dtype=np.float64)
    P, B, d = Xq.shape
    ls = state.lengthscales
    F = W.shape[0]
    val = np.empty((P, B))
    grad = np.empty((P, B, d))
    for bb in range(B):
        A = Xq[:, bb, :] / ls
        arg = A @ W.T + b                                           # [P, F]
        c = math.sqrt(2.0 * state.variance / F)
        diff = A[:, None, :] - (state.X / ls)[None, :, :]           # [P, N, d]
        r2 = np.sum(diff * diff, axis=-1)
        Kx = kernel_from_r2(state.kind, state.variance, r2)
        val[:, bb] = c * np.cos(arg) @ w[:, bb] + Kx @ v[:, bb] + state.mean_const
        dphi = -c * (np.sin(arg) * w[:, bb][None, :]) @ W / ls      # [P, d]
        dk = (2.0 * _kernel_dr2(state.kind, state.variance, r2))[:, :, None] * diff / ls
        grad[:, bb, :] = dphi + np.einsum("pnd,n->pd", dk, v[:, bb])
    return val, grad


# --------------------------------------------------------------------------------------
# A.7 sweep semantics (acquisition/optimizer.py:124-170, 247-341; sampler.py:88-123,223-273)
# --------------------------------------------------------------------------------------
def argmax_first(values: np.ndarray, axis: int = 0):
    """tf.math.argmax: first index wins ties (numpy argmax has the same rule)."""
    return np.argmax(values, axis=axis)


def argmin_first(values: np.ndarray, axis: int = 0):
    return np.argmin(values, axis=axis)


def top_k(values: np.ndarray, k: int):
    """tf.math.top_k on a 1-D array: descending values, ties by lower index first."""
    order = np.lexsort((np.arange(values.shape[0]), -values))[:k]
    return values[order], order


def ei_argmax(state: GPRState, Xq: np.ndarray, eta: float):
    vals = ei_values(state, Xq, eta)
    i = int(argmax_first(vals))
    return float(vals[i]), i


def ei_sweep_reference_shape(state: GPRState, Xq: np.ndarray, eta: float, chunk: int = 10000):
    """The reference's algorithmic shape for a large sweep: per chunk materialise K*,
    two triangular solves, column norms, EI; running arg-max (acquisition/utils.py:31-109 +
    optimizer.py:124-150).  Used as bench.py's cpu_baseline ("port")."""
    best, best_i = -np.inf, -1
    for s in range(0, Xq.shape[0], chunk):
        vals = ei_values(state, Xq[s:s + chunk], eta)
        i = int(np.argmax(vals))
        if vals[i] > best:
            best, best_i = float(vals[i]), s + i
    return best, best_i


# --------------------------------------------------------------------------------------
# A.8 objectives on [0,1]^d (reference trieste/objectives/single_objectives.py)
# --------------------------------------------------------------------------------------
def _branin_internals(x, scale, translate):  # :83-96
    x0 = x[..., 0] * 15.0 - 5.0
    x1 = x[..., 1] * 15.0
    b = 5.1 / (4 * math.pi ** 2)
    c = 5 / math.pi
    r, s, t = 6, 10, 1 / (8 * math.pi)
    return scale * ((x1 - b * x0 ** 2 + c * x0 - r) ** 2 + s * (1 - t) * np.cos(x0) + translate)


def branin(x):  # :99-107
    return _branin_internals(np.asarray(x, dtype=np.float64), 1.0, 10.0)


def scaled_branin(x):  # :110-119
    return _branin_internals(np.asarray(x, dtype=np.float64), 1 / 51.95, -44.81)


BRANIN_MINIMIZERS = (np.array([[-math.pi, 12.275], [math.pi, 2.275], [9.42478, 2.475]])
                     + np.array([5.0, 0.0])) / 15.0  # :122-131
BRANIN_MINIMUM = 0.397887
SCALED_BRANIN_MINIMUM = -1.047393

_H6_a = np.array([1.0, 1.2, 3.0, 3.2])
_H6_A = np.array([[10.0, 3.0, 17.0, 3.5, 1.7, 8.0], [0.05, 10.0, 17.0, 0.1, 8.0, 14.0],
                  [3.0, 3.5, 1.7, 10.0, 17.0, 8.0], [17.0, 8.0, 0.05, 10.0, 0.1, 14.0]])
_H6_P = np.array([[0.1312, 0.1696, 0.5569, 0.0124, 0.8283, 0.5886],
                  [0.2329, 0.4135, 0.8307, 0.3736, 0.1004, 0.9991],
                  [0.2348, 0.1451, 0.3522, 0.2883, 0.3047, 0.6650],
                  [0.4047, 0.8828, 0.8732, 0.5743, 0.1091, 0.0381]])
HARTMANN6_MINIMIZER = np.array([[0.20169, 0.150011, 0.476874, 0.275332, 0.311652, 0.6573]])
HARTMANN6_MINIMUM = -3.32237


def hartmann_6(x):  # :476-501
    x = np.asarray(x, dtype=np.float64)
    inner = -np.sum(_H6_A * (x[..., None, :] - _H6_P) ** 2, axis=-1)
    return -np.sum(_H6_a * np.exp(inner), axis=-1)


def ackley(x):  # :434-459 with 1/5 -> 1/d ("Ackley-8", "Ackley-16")
    x = (np.asarray(x, dtype=np.float64) - 0.5) * (32.768 * 2.0)
    d = x.shape[-1]
    e1 = -0.2 * np.sqrt(np.sum(x ** 2, -1) / d)
    e2 = np.sum(np.cos(2.0 * math.pi * x), -1) / d
    return -20.0 * np.exp(e1) - np.exp(e2) + 20.0 + math.e


# --------------------------------------------------------------------------------------
# Synthetic configurations shared by tests and bench.py
# --------------------------------------------------------------------------------------
def synthetic_problem(objective, d: int, N: int, seed: int = 1234):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, d))
    Yraw = objective(X)
    Y = (Yraw - Yraw.mean()) / Yraw.std()
    return X, Y


def default_lengthscales(d: int) -> np.ndarray:
    """builders.py:41, 413-423 on the unit cube: 0.2 * (upper-lower) * sqrt(d)."""
    return np.full(d, 0.2 * math.sqrt(d))
