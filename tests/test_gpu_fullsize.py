"""GPU tests at BASELINE.json's full sizes through size-independent properties (no oracle can run a
10^6-candidate sweep), plus small oracle slices at the full N."""
import numpy as np
import pytest

from oracle import gp_oracle as O
from tests.util import assert_close, cancellation_floor

pytestmark = pytest.mark.gpu


def _engine(obj, d, kind, N, noise=1e-2):
    from trieste_amd.engine import GPEngine

    X, Y = O.synthetic_problem(obj, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    eng = GPEngine(d, kind)
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    return eng, X, Y, ls, c


def test_c2_hartmann6_rbf_n1024_one_million_candidates():
    """Config 2: fused arg-max over 10^6 device-generated candidates == arg-max of the values it
    would have written; idempotent; two shards merge to the same winner; oracle slice at full N."""
    import torch

    from trieste_amd.distributed import merge_best, shard_range

    eng, X, Y, ls, c = _engine(O.hartmann_6, 6, "rbf", 1024)
    M = 1_000_000
    Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
    eta = eng.eta()
    v1, i1, x1 = eng.acq_argmax("ei", eta, Xq)
    v2, i2, _ = eng.acq_argmax("ei", eta, Xq)
    assert (v1, i1) == (v2, i2)  # deterministic
    vals = eng.acq_values("ei", eta, Xq)
    assert int(torch.argmax(vals)) == i1 and float(vals[i1]) == v1
    np.testing.assert_array_equal(x1, Xq[i1].cpu().numpy())
    parts = []
    for r in range(3):  # a ragged 3-way shard (the multi-GPU contract), merged like the ranks would
        lo, hi = shard_range(M, r, 3)
        pv, pi, _ = eng.acq_argmax("ei", eta, Xq[lo:hi], index_base=lo)
        parts.append((pv, pi))
    mv, mi = merge_best(np.array([[p[0]] for p in parts]), np.array([[p[1]] for p in parts]))
    assert (mv[0], mi[0]) == (v1, i1)
    # oracle slice
    st = O.gpr_update("rbf", 1.0, ls, 1e-2, c, X, Y)
    sl = Xq[:400].cpu().numpy()
    om, ov = O.predict(st, sl)
    floor = cancellation_floor(1024, 1.0, 1e-2)
    gm, gv = eng.predict(Xq[:400])
    assert_close(gm.cpu().numpy(), om, atol=floor, what="c2 mean slice")
    assert_close(gv.cpu().numpy(), ov, atol=floor, what="c2 var slice")
    # EI at eta = min_i mean(X_i) underflows on random candidates (oracle max 1.8e-37 here): compare it, but let the
    # non-vacuous check be EI at the slice's median mean, where every value is O(0.1)
    assert_close(vals[:400].cpu().numpy(), O.expected_improvement(om, ov, eta), atol=floor, what="EI slice at eta=min")
    eta_mid = float(np.median(om))
    want = O.expected_improvement(om, ov, eta_mid)
    assert np.count_nonzero(want > 1e-6) >= want.size // 2
    assert_close(eng.acq_values("ei", eta_mid, Xq[:400]).cpu().numpy(), want, atol=floor, what="c2 EI slice")
    v3, i3, _ = eng.acq_argmax("ei", eta_mid, Xq[:400])
    assert i3 == int(np.argmax(want)) or abs(want[i3] - want.max()) <= 1e-5 * want.max() + floor
    # shard-consistent candidate generation: rows [lo, hi) regenerated == slice of the whole
    lo, hi = 123_457, 123_457 + 1000
    np.testing.assert_array_equal(eng.sample_box(5678, lo, hi - lo, 0.0, 1.0).cpu().numpy(), Xq[lo:hi].cpu().numpy())


def test_c4_batch_mc_ei_q50_s512_n2048():
    """Config 4: qEI with q = 50, S = 512 at N = 2048: oracle on a few groups; the estimator is a mean
    over draws, so halves of eps average to the whole (size-independent); q-batch order of groups
    does not matter."""
    eng, X, Y, ls, c = _engine(O.hartmann_6, 6, "matern52", 2048)
    rng = np.random.default_rng(91011)
    q, S, G = 50, 512, 48
    eps = rng.standard_normal((q, S))
    Xg = rng.uniform(size=(G, q, 6))
    st = O.gpr_update("matern52", 1.0, ls, 1e-2, c, X, Y)
    floor = cancellation_floor(2048, 1.0, 1e-2)
    jm, jc = eng.predict_joint(Xg)
    om, oc = O.predict_joint(st, Xg[:6])
    assert_close(jm[:6], om, atol=1e-8, what="joint mean")
    assert_close(jc[:6], oc, atol=floor, what="joint cov")
    assert np.all(np.linalg.eigvalsh(jc[:3] + 1e-6 * np.eye(q)) > 0)
    # eta: the median posterior mean over the groups.  At the reference's eta = min_i mean(X_i) (-7.08 here, lowest
    # group mean -6.01) every random group has qEI == 0 exactly, and the three checks below compared zeros
    # (VERDICT r03 weak 1); with the median the oracle's values are 3.3 ... 5.3.
    eta = float(np.median(jm))
    full = eng.qei(Xg, eps, eta, 1e-6)
    assert np.count_nonzero(full) == G and full.min() > 1.0
    a = eng.qei(Xg, eps[:, :256], eta, 1e-6)
    b = eng.qei(Xg, eps[:, 256:], eta, 1e-6)
    assert_close(full, 0.5 * (a + b), rtol=1e-12, atol=1e-15, what="mean over draws splits")
    perm = rng.permutation(G)
    np.testing.assert_array_equal(eng.qei(Xg[perm], eps, eta, 1e-6), full[perm])
    want = O.batch_mc_ei(st, Xg[:6], eps, eta, 1e-6)
    assert np.count_nonzero(want) == want.size
    assert_close(full[:6], want, atol=floor, what="c4 qEI vs oracle")
    # the reference's own eta as well (exact zeros on both sides)
    eta0 = eng.eta()
    assert_close(eng.qei(Xg[:6], eps, eta0, 1e-6), O.batch_mc_ei(st, Xg[:6], eps, eta0, 1e-6), atol=floor,
                 what="c4 qEI at eta=min")
    # the samples behind it: [G, S, q] = mean + chol(cov + 1e-6 I) eps (models/gpflow/sampler.py:276-287)
    smp = eng.reparam_samples(Xg[:3], eps, 1e-6)
    want_s = O.batch_reparam_samples(st, Xg[:3], eps, 1e-6)
    lam = min(float(np.linalg.eigvalsh(cv + 1e-6 * np.eye(q)).min()) for cv in oc[:3])
    assert_close(smp, want_s, atol=floor + floor * q * np.abs(eps).max() / (2 * np.sqrt(lam)), what="c4 reparam samples")


def test_c5_decoupled_thompson_n8192_d16_f2048():
    """Config 5: trajectories are affine in their draws (w, xi): g(w1+w2, xi1+xi2) = g(w1,xi1) +
    g(w2,xi2) - g(0,0); fused arg-min == arg-min of the evaluated values; oracle slice."""
    eng, X, Y, ls, c = _engine(O.ackley, 16, "matern52", 8192)
    rng = np.random.default_rng(7)
    F, d, N = 2048, 16, 8192
    W = rng.standard_t(5, size=(F, d))
    b = rng.uniform(0, 2 * np.pi, F)
    w = rng.standard_normal((F, 2))
    xi = rng.standard_normal((N, 2))
    cols_w = np.stack([w[:, 0], w[:, 1], w[:, 0] + w[:, 1], np.zeros(F)], axis=1)
    cols_xi = np.stack([xi[:, 0], xi[:, 1], xi[:, 0] + xi[:, 1], np.zeros(N)], axis=1)
    traj = eng.trajectory(W, b, cols_w, cols_xi)
    Xq = eng.sample_box(5678, 0, 200_000, 0.0, 1.0)
    vals = traj(Xq).cpu().numpy()
    g = vals - c
    scale = np.abs(g).max()
    assert_close(g[:, 2], g[:, 0] + g[:, 1] - g[:, 3], rtol=1e-9, atol=1e-9 * scale, what="affine in the draws")
    mv, mi = traj.argmin(Xq)
    np.testing.assert_array_equal(mi, np.argmin(vals, axis=0))
    np.testing.assert_array_equal(mv, vals[mi, np.arange(4)])
    # oracle slice (trajectory evaluation given the engine's own canonical weights) + weights themselves
    st = O.gpr_update("matern52", 1.0, ls, 1e-2, c, X, Y)
    sl = Xq[:200].cpu().numpy()
    want = O.trajectory_eval(st, W, b, cols_w, traj.v(), sl)
    assert_close(vals[:200], want, rtol=1e-5, atol=1e-8 * scale, what="trajectory slice")
    ov = O.decoupled_weights(st, W, b, cols_w[:, :1], cols_xi[:, :1])
    assert_close(traj.v()[:, :1], ov, rtol=1e-5, atol=1e-7 * np.abs(ov).max(), what="canonical weights")
