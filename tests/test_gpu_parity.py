"""GPU parity tests: the HIP engine (through the C-ABI) vs the mpmath goldens and the numpy oracle.

Tolerances: 1e-5 relative (BASELINE.json north_star) plus the cancellation floor for
variance-derived quantities defined in tests/util.py.
"""
import numpy as np
import pytest

from oracle import gp_oracle as O
from tests.util import assert_close, cancellation_floor, load_goldens, load_wide_qei_goldens, reparam_sample_atol

pytestmark = pytest.mark.gpu

CASES = load_goldens()


def _engine(kind, d, variance, ls, noise, c, X, Y, variant=0):
    from trieste_amd.engine import GPEngine

    eng = GPEngine(d, kind)
    eng.set_variant(variant)
    eng.set_hyper(variance, ls, noise, c)
    eng.set_data(X, Y)
    return eng


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_engine_matches_mpmath_goldens(c):
    N, var0, noise = c["N"], c["variance"], c["noise"]
    floor = cancellation_floor(N, var0, noise)
    X, Y = np.array(c["X"]), np.array(c["Y"])
    eng = _engine(c["kind"], c["d"], var0, c["lengthscales"], noise, c["mean_const"], X, Y)
    L, W, alpha = eng.get_factor()
    assert_close(L, np.array(c["L"]), atol=floor, what="L")
    assert_close(W @ np.array(c["L"]), np.eye(N), rtol=0, atol=1e-9 * (1 + var0 / noise), what="W L = I")
    ascale = max(1.0, np.abs(np.array(c["alpha"])).max())
    assert_close(alpha, c["alpha"], atol=floor * ascale / min(noise, 1.0), what="alpha")
    Xq = np.array(c["Xq"])
    mean, var = eng.predict(Xq)
    assert_close(mean, c["mean"], atol=floor, what="mean")
    assert_close(var, c["var"], atol=floor, what="var")
    assert_close(eng.predict_mean(Xq), c["mean"], atol=floor, what="predict_mean")
    assert_close(eng.eta(), c["eta"], atol=floor, what="eta")
    if noise >= 1e-3:  # well conditioned: acquisition values end to end
        assert_close(eng.acq_values("ei", c["eta"], Xq), c["ei"], atol=floor, what="ei")
        assert_close(eng.acq_values("pi", c["eta"], Xq), c["pi"], atol=max(floor, 1e-300) * 1e3, what="pi")
        assert_close(eng.acq_values("nlcb", 1.96, Xq), c["nlcb"], atol=floor, what="nlcb")
        assert_close(eng.acq_values("aei", c["eta"], Xq), c["aei"], atol=floor, what="aei")
        assert_close(eng.qei(np.array(c["Xg"]), np.array(c["eps"]), c["eta"], c["jitter"]), c["qei"],
                     atol=floor, what="qei")
    n1, n2 = len(c["cov12"]), len(c["cov12"][0])
    assert_close(eng.cov_between(Xq[:n1], Xq[n1:n1 + n2]), c["cov12"], atol=floor, what="cov12")
    jm, jc = eng.predict_joint(np.array(c["Xg"]))
    assert_close(jm, c["joint_mean"], atol=floor, what="joint mean")
    assert_close(jc, c["joint_cov"], atol=floor, what="joint cov")
    traj = eng.trajectory(np.array(c["rff_W"]), np.array(c["rff_b"]), np.array(c["traj_w"]),
                          np.array(c["traj_xi"]))
    scale = max(1.0, np.max(np.abs(np.array(c["traj_v"]))))
    assert_close(traj.v(), c["traj_v"], atol=floor * scale / min(noise, 1.0), what="traj v")
    if noise >= 1e-3:
        assert_close(traj(Xq), c["traj"], atol=1e-7 * scale, what="trajectory")


# ----------------------------------------------------------------------------------------------
CONFIGS = [
    # (name, objective, d, kind, N, noise)
    ("branin_m52_N50", O.branin, 2, "matern52", 50, 1e-3),
    ("hartmann_rbf_N300", O.hartmann_6, 6, "rbf", 300, 1e-2),
    ("ackley8_m52_N1000", O.ackley, 8, "matern52", 1000, 1e-2),
    ("ackley8_m52_N1000_lownoise", O.ackley, 8, "matern52", 1000, 1e-5),
    ("ackley16_m32_N257", O.ackley, 16, "matern32", 257, 1e-3),
    ("ackley3_m12_N130", O.ackley, 3, "matern12", 130, 1e-3),
]


def _problem(obj, d, kind, N, noise, M=1500, seed=5678):
    X, Y = O.synthetic_problem(obj, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    st = O.gpr_update(kind, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(seed)
    Xq = rng.uniform(size=(M, d))
    Xq[:5] = X[:5]                 # exactly at training inputs (variance cancellation)
    Xq[5] = Xq[6]                  # duplicated candidate (ties -> first index)
    Xq[-3:] = 4.0 + rng.uniform(size=(3, d))  # far field: var -> variance, EI underflow
    return X, Y, ls, c, st, Xq


def _oracle_argmax_agrees(idx, oracle_vals, tol):
    """The engine's arg-max index must be the oracle's whenever the oracle's top-2 gap exceeds the tolerance; inside
    the tolerance band any index whose oracle value is within it is a correct answer of equal standing."""
    oi = int(np.argmax(oracle_vals))
    if idx == oi:
        return True
    return abs(oracle_vals[oi] - oracle_vals[idx]) <= tol


@pytest.mark.parametrize("variant", [0, 1, 2, 9], ids=["default-policy", "fused-dma", "rowsplit", "fused-regstage"])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_sweep_matches_oracle(cfg, variant):
    """The sweep kernel under every launch policy: the default (row-group split for launches with few candidate
    blocks), the fused form forced (tgp_set_variant bit 0: what large launches use -- the LDS-DMA kernel for dp <= 16, or the
    register-staged one with bit 3) and the split forced (bit 1)."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise)
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y, variant)
    mean, var = eng.predict(Xq)
    om, ov = O.predict(st, Xq)
    assert_close(mean, om, atol=floor * 10, what="mean")
    assert_close(var, ov, atol=floor, what="var")
    eta = eng.eta()
    assert_close(eta, O.eta_min_mean(st), atol=floor, what="eta")
    ei = eng.acq_values("ei", eta, Xq)
    oei = O.expected_improvement(om, ov, eta)
    assert_close(ei, oei, atol=floor, what="ei")
    # fused arg-max == arg-max of the engine's own values (first index on ties); its value agrees with the
    # oracle's maximum and its INDEX with the oracle's arg-max (up to the tolerance band)
    val, idx, x = eng.acq_argmax("ei", eta, Xq)
    assert idx == int(np.argmax(ei)) and val == ei[idx]
    assert_close(val, np.max(oei), atol=floor, what="max ei")
    assert _oracle_argmax_agrees(idx, oei, 1e-5 * np.max(oei) + floor), (idx, int(np.argmax(oei)))
    np.testing.assert_array_equal(x, Xq[idx])
    # top-k == stable descending sort of the engine's values
    k = 17
    tv, ti = eng.acq_topk("ei", eta, Xq, k)
    ov_, oi_ = O.top_k(ei, k)
    np.testing.assert_array_equal(ti, oi_)
    np.testing.assert_array_equal(tv, ov_)


@pytest.mark.parametrize("M", [1, 7, 64, 65, 128, 1000, 2048])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_predict_at_a_handful_of_points_matches_the_oracle_and_the_sweep(cfg, M):
    """Round 6: tgp_predict at <= 2048 points is K*^T, one skinny triangular product W K* and a two-pass tail (the arrays of the
    value-and-gradient call) instead of a sweep launch -- mean and variance against the oracle at the sweep's tolerances, against
    the sweep itself (tgp_set_variant bit 10) to rounding, mean-only and variance-only calls, training inputs, duplicates and the
    far field among the points; 2049 points go through the sweep again (the same values as with bit 10: bit for bit)."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq_all = _problem(obj, d, kind, N, noise, M=2100)
    Xq = np.ascontiguousarray(np.concatenate([Xq_all[:M - 3], Xq_all[-3:]]) if M >= 7 else Xq_all[:M])
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y, 0)
    sweep = _engine(kind, d, 1.0, ls, noise, c, X, Y, 1024)
    mean, var = eng.predict(Xq)
    om, ov = O.predict(st, Xq)
    assert_close(mean, om, atol=floor * 10, what="mean at a handful of points")
    assert_close(var, ov, atol=floor, what="var at a handful of points")
    ms, vs = sweep.predict(Xq)
    assert_close(mean, ms, atol=floor * 10, what="mean: skinny product vs sweep")
    assert_close(var, vs, atol=floor, what="var: skinny product vs sweep")
    assert np.all(var >= 1e-12)
    np.testing.assert_array_equal(eng.predict_mean(Xq), sweep.predict_mean(Xq))   # (its own kernel either way)
    big = np.ascontiguousarray(Xq_all[:2049])
    for a, b in zip(eng.predict(big), sweep.predict(big)):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[3], CONFIGS[1]], ids=lambda c: c[0])
def test_launch_policies_agree_to_rounding(cfg):
    """Fused and row-group-split launches form the same products; only the order in which the row blocks' partial
    column norms / mean terms are added differs (one chain vs per-group partials + combine): they agree to a few
    ulps of the accumulated sums, far below the parity tolerance, and pick the same arg-max."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=700)
    outs = []
    for variant in (1, 2, 9):
        eng = _engine(kind, d, 1.0, ls, noise, c, X, Y, variant)
        m, v = eng.predict(Xq)
        outs.append((np.asarray(m), np.asarray(v), eng.acq_argmax("ei", eng.eta(), Xq)[:2]))
    assert_close(outs[0][1], outs[1][1], rtol=0, atol=1e-13, what="var")
    assert_close(outs[0][0], outs[1][0], rtol=0, atol=1e-13 * max(1.0, np.abs(outs[0][0]).max()), what="mean")
    assert outs[0][2][1] == outs[1][2][1]
    # the two fused kernels (LDS-DMA staging / register staging) sum in the same order: identical bits
    np.testing.assert_array_equal(outs[0][1], outs[2][1])
    np.testing.assert_array_equal(outs[0][0], outs[2][0])
    assert outs[0][2] == outs[2][2]


def test_ties_pick_first_index_and_sharding_is_consistent():
    _, obj, d, kind, N, noise = CONFIGS[2]
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=1000)
    Xq = np.concatenate([Xq, Xq[::-1]], axis=0)  # every candidate appears twice
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    eta = eng.eta()
    val, idx, _ = eng.acq_argmax("ei", eta, Xq)
    ei = eng.acq_values("ei", eta, Xq)
    assert idx == int(np.argmax(ei))
    twin = len(Xq) - 1 - idx
    assert ei[twin] == ei[idx] and idx < twin
    # contiguous shards with index_base reproduce the global winner (the multi-GPU contract)
    best = (-np.inf, -1)
    for lo, hi in ((0, 700), (700, 1300), (1300, 2000)):
        v, i, _ = eng.acq_argmax("ei", eta, Xq[lo:hi], index_base=lo)
        if v > best[0] or (v == best[0] and i < best[1]):
            best = (v, i)
    assert best == (val, idx)


def test_device_resident_inputs_match_host_inputs():
    import torch

    _, obj, d, kind, N, noise = CONFIGS[1]
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=777)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    m0, v0 = eng.predict(Xq)
    Xd = torch.from_numpy(Xq).cuda()
    eng.use_torch_stream()
    m1, v1 = eng.predict(Xd)
    assert m1.is_cuda and v1.is_cuda
    np.testing.assert_array_equal(m1.cpu().numpy(), m0)
    np.testing.assert_array_equal(v1.cpu().numpy(), v0)
    eta = eng.eta()
    a = eng.acq_argmax("ei", eta, Xq)
    b = eng.acq_argmax("ei", eta, Xd)
    assert a[0] == b[0] and a[1] == b[1]
    np.testing.assert_array_equal(a[2], b[2])
    ms, n = eng.last_kernel_ms()
    assert ms > 0 and n == 1


def test_edge_cases_and_errors():
    from trieste_amd._lib import NotPositiveDefiniteError
    from trieste_amd.engine import GPEngine

    eng = GPEngine(2, "matern52")
    with pytest.raises(RuntimeError):
        eng.set_data(np.zeros((3, 2)), np.zeros(3))  # hyper-parameters not set
    eng.set_hyper(1.0, [0.3, 0.4], 1e-2, 0.1)
    with pytest.raises(RuntimeError):
        eng.predict(np.zeros((1, 2)))  # no data yet
    with pytest.raises(ValueError):
        eng.set_data(np.zeros((3, 3)), np.zeros(3))  # wrong dimension (reference: ValueError)
    with pytest.raises(ValueError):
        eng.set_data(np.zeros((3, 2)), np.zeros(4))
    # N = 1
    eng.set_data(np.array([[0.2, 0.7]]), np.array([1.5]))
    st = O.gpr_update("matern52", 1.0, np.array([0.3, 0.4]), 1e-2, 0.1, np.array([[0.2, 0.7]]), np.array([1.5]))
    Xq = np.random.default_rng(1).uniform(size=(131, 2))
    m, v = eng.predict(Xq)
    om, ov = O.predict(st, Xq)
    assert_close(m, om, atol=1e-12, what="N=1 mean")
    assert_close(v, ov, atol=1e-12, what="N=1 var")
    # empty query
    m, v = eng.predict(np.zeros((0, 2)))
    assert m.shape == (0,) and v.shape == (0,)
    with pytest.raises(ValueError):
        eng.acq_argmax("ei", 0.0, np.zeros((0, 2)))
    # single query point
    m, v = eng.predict(Xq[:1])
    assert_close(m, om[:1], atol=1e-12)
    # not positive definite: duplicated inputs with (numerically) no noise
    eng.set_hyper(1.0, [0.3, 0.4], 1e-30, 0.0)
    with pytest.raises(NotPositiveDefiniteError):
        eng.set_data(np.array([[0.5, 0.5], [0.5, 0.5], [0.1, 0.2]]), np.array([1.0, 1.0, 0.0]))
    with pytest.raises(RuntimeError):
        eng.predict(Xq)  # failed update leaves the model unusable until the next good update
    # update with new data of a different size (reference models.py:146-165: dynamic shapes)
    eng.set_hyper(1.0, [0.3, 0.4], 1e-2, 0.0)
    X2 = np.random.default_rng(2).uniform(size=(200, 2))
    Y2 = O.branin(X2)
    Y2 = (Y2 - Y2.mean()) / Y2.std()
    eng.set_data(X2, Y2)
    st2 = O.gpr_update("matern52", 1.0, np.array([0.3, 0.4]), 1e-2, 0.0, X2, Y2)
    m, v = eng.predict(Xq)
    om, ov = O.predict(st2, Xq)
    assert_close(m, om, atol=1e-9)
    assert_close(v, ov, atol=1e-10)


_sample_atol = reparam_sample_atol

WIDE = load_wide_qei_goldens()


@pytest.mark.parametrize("variant", [0, 1024, 4], ids=["skinny-product", "packed-128x256", "slots-256x128"])
@pytest.mark.parametrize("c", WIDE, ids=[c["name"] for c in WIDE])
def test_engine_wide_qei_matches_mpmath_goldens(c, variant):
    """The engine against 50-digit arithmetic at q = 9, 17, 33, 50 -- every instantiation of the one-wave qEI tail
    (QP = 16, 32, 64) and BASELINE config 4's group size -- joint mean / covariance, the reparametrised samples
    (tgp_reparam_samples) and qEI (tgp_qei) at an incumbent where every value is O(1).  Independent of the numpy oracle."""
    eng = _engine(c["kind"], c["d"], c["variance"], c["lengthscales"], c["noise"], c["mean_const"], np.array(c["X"]),
                  np.array(c["Y"]), variant)
    floor = cancellation_floor(c["N"], c["variance"], c["noise"])
    Xg, eps = np.array(c["Xg"]), np.array(c["eps"])
    jm, jc = eng.predict_joint(Xg)
    assert_close(jm, c["joint_mean"], atol=floor * 10, what="wide joint mean")
    if "joint_cov" in c:
        assert_close(jc, c["joint_cov"], atol=floor, what="wide joint cov")
    atol = reparam_sample_atol(jc, floor, eps)
    assert_close(eng.reparam_samples(Xg, eps, c["jitter"]), c["samples"], atol=atol, what=f"golden samples q={c['q']}")
    want = np.array(c["qei"])
    assert np.all(want > 0.1)
    assert_close(eng.qei(Xg, eps, c["eta"], c["jitter"]), want, atol=atol, what=f"golden qEI q={c['q']}")


@pytest.mark.parametrize("variant", [0, 1024, 4], ids=["skinny-product", "packed-128x256", "slots-256x128"])
@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[2], CONFIGS[4]], ids=lambda c: c[0])
def test_joint_and_qei_match_oracle(cfg, variant):
    """(Round 6: a call of <= 2048 points takes the skinny-product form of tgp_joint_forward by default; variant bit 10 keeps the
    joint kernel for it, so that every case below still runs on all three.)  Joint mode on both kernels: contiguously packed groups in 128 x 256 tiles with the LDS Gram phase (default,
    dp <= 16) and the first-generation 64-column slots (variant bit 2).  Group counts around the block capacity
    (floor(256 / q) groups per block), q from 1 to 64, a group starting at a training input."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, _ = _problem(obj, d, kind, N, noise)
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y, variant)
    rng = np.random.default_rng(7)
    for q, G, S in ((1, 9, 8), (1, 600, 4), (3, 50, 16), (3, 171, 4), (5, 11, 32), (50, 7, 64), (50, 5, 8), (64, 3, 8),
                    (64, 9, 4), (17, 6, 8), (17, 31, 4), (33, 15, 4),
                    # the qEI tail's template bounds (8 / 16 / 32 / 64 base draws in registers) and ragged sample sets
                    (8, 5, 70), (9, 4, 8), (16, 3, 130), (32, 3, 65)):
        Xg = rng.uniform(size=(G, q, d))
        Xg[0, 0] = X[0]
        jm, jc = eng.predict_joint(Xg)
        om, oc = O.predict_joint(st, Xg)
        assert_close(jm, om, atol=floor, what=f"joint mean q={q}")
        assert_close(jc, oc, atol=floor, what=f"joint cov q={q}")
        eps = rng.normal(size=(q, S))
        # eta = the median posterior mean over the groups, NOT the training minimum: with eta = min_i mean(X_i)
        # every random group has qEI = max(eta - min, 0) = 0 exactly and the comparison is 0 == 0 (VERDICT r03
        # weak 1) -- a wrong Cholesky row or a dropped eps column would pass.
        eta = float(np.median(om))
        got = eng.qei(Xg, eps, eta, 1e-6)
        want = O.batch_mc_ei(st, Xg, eps, eta, 1e-6)
        assert np.count_nonzero(want) >= want.size // 2, f"vacuous qEI comparison at q={q}: {want}"
        assert_close(got, want, atol=floor, what=f"qei q={q}")
        # ... and at the reference's own eta (function.py:1135-1147), where most values are exactly zero
        eta0 = O.eta_min_mean(st)
        assert_close(eng.qei(Xg, eps, eta0, 1e-6), O.batch_mc_ei(st, Xg, eps, eta0, 1e-6), atol=floor,
                     what=f"qei at eta=min q={q}")
        # the samples themselves (the tail's Cholesky + sample code without the max / mean reduction that could
        # hide an error): BatchReparametrizationSampler.sample, models/gpflow/sampler.py:276-287
        if (q, G) in ((9, 4), (16, 3), (17, 6), (32, 3), (50, 5), (64, 3), (5, 11)):
            smp = eng.reparam_samples(Xg, eps, 1e-6)
            want_s = O.batch_reparam_samples(st, Xg, eps, 1e-6)
            assert_close(smp, want_s, atol=_sample_atol(oc, floor, eps), what=f"reparam samples q={q}")


def test_trajectories_match_oracle_and_argmin():
    _, obj, d, kind, N, noise = CONFIGS[2]
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=900)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(11)
    F, B = 96, 3
    W = rng.standard_t(5, size=(F, d))
    b = rng.uniform(0, 2 * np.pi, size=F)
    w = rng.normal(size=(F, B))
    xi = rng.normal(size=(N, B))
    traj = eng.trajectory(W, b, w, xi)
    ov = O.decoupled_weights(st, W, b, w, xi)
    vs = np.abs(ov).max()
    assert_close(traj.v(), ov, rtol=1e-5, atol=1e-7 * vs, what="v")
    got = traj(Xq)
    want = O.trajectory_eval(st, W, b, w, traj.v(), Xq)
    assert_close(got, want, rtol=1e-5, atol=1e-8 * max(1.0, vs), what="traj eval")
    vals, idx = traj.argmin(Xq)
    np.testing.assert_array_equal(idx, np.argmin(got, axis=0))
    np.testing.assert_array_equal(vals, got[idx, np.arange(B)])
    # per-trajectory inputs [M, B, d]
    Xb = rng.uniform(size=(50, B, d))
    got_b = traj(Xb)
    want_b = O.trajectory_eval(st, W, b, w, traj.v(), Xb)
    assert_close(got_b, want_b, rtol=1e-5, atol=1e-8 * max(1.0, vs), what="traj eval per-traj inputs")


def test_sample_box_is_uniform_and_shard_consistent():
    from trieste_amd.engine import GPEngine

    eng = GPEngine(3, "rbf")
    lo, up = np.array([0.0, -1.0, 2.0]), np.array([1.0, 1.0, 5.0])
    a = eng.sample_box(42, 0, 10000, lo, up).cpu().numpy()
    assert a.shape == (10000, 3)
    assert np.all(a >= lo) and np.all(a < up)
    assert_close(a.mean(0), (lo + up) / 2, rtol=0, atol=0.05)
    b = eng.sample_box(42, 6000, 4000, lo, up).cpu().numpy()
    np.testing.assert_array_equal(a[6000:], b)
    c = eng.sample_box(43, 0, 100, lo, up).cpu().numpy()
    assert not np.array_equal(a[:100], c)


# ---- full-size, size-independent properties (no oracle at this size) --------------------------
def test_headline_size_properties():
    """N = 4096, d = 8, Matern-5/2 (BASELINE headline): identities that hold at any size."""
    d, N, noise = 8, 4096, 1e-2
    X, Y = O.synthetic_problem(O.ackley, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    eng = _engine("matern52", d, 1.0, ls, noise, c, X, Y)
    L, W, alpha = eng.get_factor()
    # (1) W is the inverse of L, and L L^T reproduces K + noise I on a random probe
    rng = np.random.default_rng(3)
    z = rng.normal(size=N)
    assert_close(W @ (L @ z), z, rtol=0, atol=1e-8, what="W L z = z")
    K = O.kernel_matrix("matern52", 1.0, ls, X)
    K[np.diag_indices(N)] += noise
    assert_close(L @ (L.T @ z), K @ z, rtol=1e-9, atol=1e-9, what="L L^T z = K z")
    # (2) at the training inputs: mean = Y - noise * alpha (exact identity), var = noise-ish small
    m, v = eng.predict(X[:512])
    assert_close(m, Y[:512] - noise * alpha[:512], rtol=1e-7, atol=1e-8, what="mean at train")
    assert np.all(v > 0) and np.all(v < noise * 1.0001)
    # (3) a small oracle slice: 300 candidates against the (slow) CPU restatement
    st = O.gpr_update("matern52", 1.0, ls, noise, c, X, Y)
    Xq = rng.uniform(size=(300, d))
    om, ov = O.predict(st, Xq)
    gm, gv = eng.predict(Xq)
    floor = cancellation_floor(N, 1.0, noise)
    assert_close(gm, om, atol=floor, what="mean vs oracle")
    assert_close(gv, ov, atol=floor, what="var vs oracle")
    # (4) far field: var -> variance, mean -> c, EI underflows to exactly 0 (tf semantics)
    far = 10.0 + rng.uniform(size=(4, d))
    fm, fv = eng.predict(far)
    assert_close(fm, np.full(4, c), rtol=0, atol=1e-12)
    assert_close(fv, np.ones(4), rtol=0, atol=1e-12)


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_acq_value_and_gradient_match_oracle(cfg):
    """tgp_acq_value_grad vs the oracle's analytic gradient (itself checked against central finite
    differences in tests/test_oracle_gradient.py): EI, PI, -LCB; includes a point at a training
    input with tiny noise (clipped variance -> zero variance-gradient) and far-field points."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=70)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    eta = eng.eta()
    for acq, par in (("ei", eta), ("pi", eta), ("nlcb", 1.96), ("aei", eta)):
        val, grad = eng.acq_value_grad(acq, par, Xq)
        oval, ograd = O.acq_value_and_grad(st, acq, par, Xq)
        floor = cancellation_floor(N, 1.0, noise)
        assert_close(val, oval, atol=floor, what=f"{acq} value")
        # gradients of the variance inherit the cancellation floor times |d k / dx| ~ 1 / lengthscale
        gscale = np.abs(ograd).max() + 1e-300
        assert_close(grad, ograd, rtol=1e-5, atol=max(floor * 1e3, 1e-9 * gscale), what=f"{acq} gradient")
        vals2 = eng.acq_values(acq, par, Xq)
        assert_close(val, vals2, rtol=1e-9, atol=floor, what=f"{acq} value == sweep value")


def _dense_joint_vjp(st, Xq, gmean, gcov):
    """d/dXq of sum gmean * mean + sum gcov * cov on dense numpy arrays (explicit solves against the oracle's factor): the
    checker of tgp_joint_vjp."""
    G, q, d = Xq.shape
    ls = st.lengthscales
    alpha = O._solve_triangular(st.L.T, O._solve_triangular(st.L, st.err, lower=True), lower=False)
    out = np.zeros((G, q, d))
    for g in range(G):
        Gs = gcov[g] + gcov[g].T
        diff = (Xq[g][:, None, :] - st.X[None, :, :]) / ls
        r2 = np.sum(diff * diff, -1)
        Kq = O.kernel_from_r2(st.kind, st.variance, r2)
        dk = (2.0 * O._kernel_dr2(st.kind, st.variance, r2))[:, :, None] * diff / ls
        T = O._solve_triangular(st.L.T, O._solve_triangular(st.L, (Gs @ Kq).T, lower=True), lower=False).T   # (Gs Kq) K^-1
        out[g] = np.einsum("qnd,qn->qd", dk, gmean[g][:, None] * alpha[None, :] - T)
        dq = (Xq[g][:, None, :] - Xq[g][None, :, :]) / ls
        dkq = (2.0 * O._kernel_dr2(st.kind, st.variance, np.sum(dq * dq, -1)))[:, :, None] * dq / ls
        out[g] += np.einsum("ij,ijd->id", Gs * (1.0 - np.eye(q)), dkq)
    return out


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_joint_forward_and_vjp_of_a_handful_of_batches_match_the_oracle(cfg):
    """Round 6: tgp_joint_forward (predict_joint of the few q-batches an L-BFGS-B iteration holds, as a skinny product) against the
    oracle's predict_joint AND the joint kernel's (tgp_predict_joint), tgp_joint_vjp against a dense numpy evaluation of the same
    vector-Jacobian product with random, NON-symmetric adjoints -- q from 1 to 64, group counts 1 ... 300, a group starting at
    a training input, a group with two nearly coincident points, 2048 points exactly; more than 2048 points is a shape error."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, _ = _problem(obj, d, kind, N, noise)
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(17)
    for q, G in ((1, 1), (1, 70), (3, 21), (5, 300), (17, 6), (50, 10), (64, 3), (64, 32)):
        Xg = rng.uniform(size=(G, q, d))
        Xg[0, 0] = X[0]
        if q >= 3:
            Xg[-1, 1] = Xg[-1, 0] + 1e-4
        jm, jc = eng.joint_forward(Xg)
        om, oc = O.predict_joint(st, Xg)
        assert_close(jm, om, atol=floor * 10, what=f"joint_forward mean q={q} G={G}")
        # (off-diagonal entries are differences of O(1) numbers formed by a k-split product: tgp_cov_between's tolerance,
        # the same pipeline; the diagonal holds the plain floor)
        assert_close(jc, oc, atol=floor * 10, what=f"joint_forward cov q={q} G={G}")
        dg = np.arange(q)
        assert_close(jc[:, dg, dg], oc[:, dg, dg], atol=floor, what=f"joint_forward variances q={q} G={G}")
        np.testing.assert_array_equal(jc, np.swapaxes(jc, -1, -2))   # both triangles from one entry of the product
        if q <= 64:
            eng.set_variant(1024)   # the joint kernel itself (a call this small takes the skinny product by default)
            km, kc = eng.predict_joint(Xg)
            eng.set_variant(0)
            assert_close(jm, km, atol=floor * 10, what="joint_forward mean vs the joint kernel")
            assert_close(jc, kc, atol=floor * 10, what="joint_forward cov vs the joint kernel")
        gm, gc = rng.normal(size=(G, q)), rng.normal(size=(G, q, q))
        got = eng.joint_vjp(Xg, gm, gc)
        want = _dense_joint_vjp(st, Xg, gm, gc)
        gscale = np.abs(want).max() + 1e-300
        assert_close(got, want, rtol=1e-5, atol=max(floor * 1e3 * q, 1e-9 * gscale), what=f"joint_vjp q={q} G={G}")
        np.testing.assert_array_equal(eng.joint_vjp(Xg, gm, gc), got)   # a fixed summation order: bit-identical call to call
    # joint mode is float64 whatever the sweep precision: a small call takes the same skinny product under "auto"
    eng.set_precision("auto")
    am, ac = eng.predict_joint(Xg)
    eng.set_precision("f64")
    fm, fc = eng.predict_joint(Xg)
    np.testing.assert_array_equal(am, fm)
    np.testing.assert_array_equal(ac, fc)
    np.testing.assert_array_equal(fc, jc)   # ... which is tgp_joint_forward's arithmetic
    # device-resident arguments (torch tensors on the GPU): the same bits as the host-buffer call
    import torch
    Xd, gmd, gcd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (Xg, gm, gc))
    jmd, jcd = eng.joint_forward(Xd)
    np.testing.assert_array_equal(jmd.cpu().numpy(), jm)
    np.testing.assert_array_equal(jcd.cpu().numpy(), jc)
    np.testing.assert_array_equal(eng.joint_vjp(Xd, gmd, gcd).cpu().numpy(), got)
    with pytest.raises(ValueError):
        eng.joint_vjp(Xd, gm, gc)   # mixed residency
    with pytest.raises(ValueError):
        eng.joint_forward(rng.uniform(size=(41, 50, d)))
    from trieste_amd import _lib
    big = np.ascontiguousarray(rng.uniform(size=(2049, 1, d)))
    out = np.empty(2049 * (1 + 1))
    rc = eng._lib.tgp_joint_forward(eng._h, big.ctypes.data, 2049, 1, out.ctypes.data, out[2049:].ctypes.data, _lib.HOST)
    assert rc == _lib.TGP_ERR_SHAPE


@pytest.mark.parametrize("cfg", [CONFIGS[0], CONFIGS[1], CONFIGS[2], CONFIGS[4]], ids=lambda c: c[0])
def test_qei_value_and_gradient_match_the_oracle(cfg, monkeypatch):
    """The gradient of BatchMonteCarloExpectedImprovement w.r.t. the batch points as the host layer assembles it (engine:
    joint_forward + joint_vjp; host: q x q factorisations, sample reduction, Cholesky adjoint) against the oracle's FORWARD-mode
    derivative of the reference's computation (predict_joint -> cholesky -> reparametrised samples -> mean of max(eta - min, 0):
    function.py:1181-1186, sampler.py:276-287), at an incumbent where most values are non-zero."""
    from trieste_amd.acquisition.function import batch_monte_carlo_expected_improvement

    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, _ = _problem(obj, d, kind, N, noise)
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(23)
    for q, G, S in ((1, 6, 32), (3, 5, 64), (7, 3, 48), (9, 40, 70), (17, 4, 130), (33, 2, 64), (50, 2, 96), (64, 2, 33)):
        Xg = rng.uniform(size=(G, q, d))
        eps = rng.normal(size=(q, S))
        eta = float(np.median(O.predict_joint(st, Xg)[0]))

        class _Sampler:
            def eps(self, qq):
                assert qq == q
                return eps

        fn = batch_monte_carlo_expected_improvement.__new__(batch_monte_carlo_expected_improvement)
        fn._engine, fn._sampler, fn._eta, fn._jitter, fn._sample_size = eng, _Sampler(), eta, 1e-6, S
        val, grad = fn.value_and_gradient(Xg)                      # tgp_qei_value_grad: the q x q arithmetic on the device too
        assert eng.qei_value_grad_fits(q, S)
        np.testing.assert_array_equal(fn.value_and_gradient(Xg)[1], grad)   # no atomics: bit-identical call to call
        monkeypatch.setattr(type(eng), "qei_value_grad_fits", staticmethod(lambda q_, S_: False))
        hval, hgrad = fn.value_and_gradient(Xg)                    # tgp_joint_forward + the HOST's adjoint + tgp_joint_vjp
        monkeypatch.undo()
        oval, ograd = O.batch_mc_ei_value_and_grad(st, Xg, eps, eta, 1e-6)
        assert_close(hval, oval, atol=floor, what=f"qEI value (host adjoint) q={q}")
        assert_close(hgrad, ograd, rtol=1e-5, atol=max(floor * 1e3 * q, 1e-7 * (np.abs(ograd).max() + 1e-300)),
                     what=f"qEI gradient (host adjoint) q={q}")
        assert np.count_nonzero(oval) >= oval.size // 2, f"vacuous qEI comparison at q={q}: {oval}"
        assert_close(val, oval, atol=floor, what=f"qEI value q={q}")
        assert_close(val, eng.qei(Xg, eps, eta, 1e-6), atol=floor, what=f"qEI value == tgp_qei q={q}")
        gscale = np.abs(ograd).max() + 1e-300
        assert_close(grad, ograd, rtol=1e-5, atol=max(floor * 1e3 * q, 1e-7 * gscale), what=f"qEI gradient q={q}")
    assert not eng.qei_value_grad_fits(64, 30000) and not eng.qei_value_grad_fits(65, 8)
    with pytest.raises(ValueError):   # the tail's LDS holds 8 (2 q (q|1) + 128) + 4 S bytes: the shape error of the C-ABI, not a fault
        eng.qei_value_grad(rng.uniform(size=(2, 64, d)), rng.normal(size=(64, 30000)), 0.0, 1e-6)


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_nlml_value_and_gradient_match_oracle(cfg):
    """tgp_nlml vs the oracle (whose gradient is finite-difference checked in test_oracle_gradient.py)."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, _ = _problem(obj, d, kind, N, noise, M=8)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    val, grad = eng.nlml()
    oval, ograd = O.nlml_and_grad(st)
    assert_close(val, oval, rtol=1e-9, atol=1e-7, what="nlml")
    vonly, gnone = eng.nlml(with_gradient=False)
    assert gnone is None and vonly == val  # value-only path: same reduction, no K^-1 product
    assert_close(grad, ograd, rtol=1e-5, atol=1e-7 * np.abs(ograd).max() + 1e-6 / noise * 1e-6, what="nlml gradient")


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_covariance_between_points_matches_oracle(cfg):
    """tgp_cov_between (models.py:188-254) vs the oracle at ragged sizes (1, 63, 64, 65, 130 points),
    its transpose symmetry, and its diagonal against predict's unclipped variance."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=200)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    floor = cancellation_floor(N, 1.0, noise)
    for p1, p2 in ((1, 1), (63, 65), (64, 130), (5, 200)):
        X1, X2 = Xq[:p1], Xq[200 - p2:]
        cov = eng.cov_between(X1, X2)
        assert cov.shape == (p1, p2)
        assert_close(cov, O.covariance_between_points(st, X1, X2), atol=floor * 10, what=f"cov {p1}x{p2}")
        assert_close(eng.cov_between(X2, X1), cov.T, rtol=1e-12, atol=floor, what="cov symmetry")
    _, var_raw = O.predict(st, Xq[:70], clip=False)
    assert_close(np.diag(eng.cov_between(Xq[:70], Xq[:70])), var_raw, atol=floor, what="diag == raw variance")
    # training inputs with themselves: K - K (K + noise I)^-1 K, tiny for tiny noise
    Xt = X[:33]
    assert_close(eng.cov_between(Xt, Xt), O.covariance_between_points(st, Xt, Xt), atol=floor, what="cov at data")


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_trajectory_value_and_gradient_match_oracle(cfg):
    """tgp_traj_value_grad vs the oracle's analytic gradient (finite-difference checked in
    tests/test_oracle_gradient.py), per-trajectory inputs [P, B, d]; values also vs tgp_traj_eval."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, _ = _problem(obj, d, kind, N, noise, M=8)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(99)
    F, B, P = 300, 3, 37
    from trieste_amd.sampler import sample_rff_basis

    W, b = sample_rff_basis(kind, F, d, rng)
    w, xi = rng.standard_normal((F, B)), rng.standard_normal((N, B))
    traj = eng.trajectory(W, b, w, xi)
    v = traj.v()
    Xp = rng.uniform(size=(P, B, d))
    Xp[0, 0] = X[3]  # at a training input (r = 0: Matern-1/2 has a kink there; value still defined)
    val, grad = traj.value_and_gradient(Xp)
    oval, ograd = O.trajectory_value_and_grad(st, W, b, w, v, Xp)
    scale = max(1.0, np.abs(v).max())
    assert_close(val, oval, rtol=1e-7, atol=1e-8 * scale, what="trajectory value")
    assert_close(val, traj(Xp), rtol=1e-10, atol=1e-9 * scale, what="value == traj_eval")
    gs = np.abs(ograd).max()
    skip0 = kind == "matern12"  # d k / dx is discontinuous at r = 0 for Matern-1/2
    sl = slice(1, None) if skip0 else slice(None)
    assert_close(grad[sl], ograd[sl], rtol=1e-6, atol=1e-8 * gs, what="trajectory gradient")


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_exact_joint_samples_match_oracle(cfg):
    """tgp_sample_joint (interface.py:135-137 -> gpflow predict_f_samples) vs the oracle for ragged
    point counts (1, 64, 65, 300: the n x n factorisation runs on the device), including exact
    duplicates of training inputs (covariance ~ 0 there: the jitter carries the factorisation)."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=300)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(17)
    floor = cancellation_floor(N, 1.0, noise)
    for n, S in ((1, 3), (64, 5), (65, 70), (300, 9)):
        eps = rng.standard_normal((n, S))
        got = eng.sample_joint(Xq[:n], eps, 1e-6)
        assert got.shape == (S, n)
        ref = O.joint_samples(st, Xq[:n], eps, 1e-6)
        # the factor of (cov + 1e-6 I) amplifies covariance error by at most ~1 / sqrt(jitter)
        assert_close(got, ref, rtol=1e-5, atol=max(floor * 1e3, 1e-9) * 30, what=f"joint samples n={n}")
    import torch

    eps = rng.standard_normal((130, 4))
    dev = eng.sample_joint(torch.from_numpy(Xq[:130]).cuda(), torch.from_numpy(eps).cuda(), 1e-6)
    assert_close(dev.cpu().numpy(), eng.sample_joint(Xq[:130], eps, 1e-6), rtol=0, atol=0, what="device == host inputs")
    with pytest.raises(ValueError):
        eng.sample_joint(Xq[:5], eps[:4], 1e-6)


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_rff_weight_posterior_trajectories_match_oracle(cfg):
    """tgp_traj_create_rff (sampler.py:518-591) vs the oracle in design space (F < N) and gram space
    (N <= F): weights theta, trajectory values (shared and per-trajectory inputs), fused arg-min and
    gradients."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=150)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    from trieste_amd.sampler import sample_rff_basis

    for F in (max(8, N // 3), N + 37):
        rng = np.random.default_rng(F)
        W, b = sample_rff_basis(kind, F, d, rng)
        B = 3
        eps = rng.standard_normal((F, B))
        traj = eng.trajectory_rff(W, b, eps)
        oth = O.rff_theta(st, W, b, eps)
        tscale = max(1.0, np.abs(oth).max())
        # both factorisations amplify rounding by cond ~ |Phi|^2 / noise
        assert_close(traj.theta(), oth, rtol=1e-6, atol=1e-9 * tscale / min(noise, 1.0), what=f"theta F={F}")
        ref = O.rff_trajectory_eval(st, W, b, traj.theta(), Xq)
        assert_close(traj(Xq), ref, rtol=1e-9, atol=1e-9 * tscale, what="rff trajectory (shared inputs)")
        Xp = rng.uniform(size=(21, B, d))
        assert_close(traj(Xp), O.rff_trajectory_eval(st, W, b, traj.theta(), Xp), rtol=1e-9, atol=1e-9 * tscale,
                     what="rff trajectory (per-trajectory inputs)")
        v, i = traj.argmin(Xq)
        np.testing.assert_array_equal(i, np.argmin(ref, axis=0))
        val, grad = traj.value_and_gradient(Xp)
        oval, ograd = O.trajectory_value_and_grad(st, W, b, traj.theta(), np.zeros((N, B)), Xp)
        assert_close(val, oval, rtol=1e-9, atol=1e-9 * tscale, what="rff value")
        assert_close(grad, ograd, rtol=1e-7, atol=1e-8 * np.abs(ograd).max(), what="rff gradient")
        with pytest.raises(RuntimeError):
            traj.v()


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_append_data_equals_full_refactorisation(cfg):
    """tgp_append_data (rank-k fast path of `update`) vs tgp_set_data on the concatenated data and vs the
    oracle: factor, alpha, eta, posterior; k = 1, 7 and 70 appended rows, chained appends, crossing a
    64-row block boundary and (ackley N=257 -> 512 padding stays, N=1000 + 70 crosses 1024) the padded size."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=120)
    rng = np.random.default_rng(23)
    floor = cancellation_floor(N + 80, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    Xall, Yall = X, Y
    for k in (1, 7, 70):
        Xn = rng.uniform(size=(k, d))
        Yn = rng.standard_normal(k) * 0.3 + c
        eng.append_data(Xn, Yn)
        Xall, Yall = np.concatenate([Xall, Xn]), np.concatenate([Yall, Yn])
        assert eng.N == Xall.shape[0]
        full = _engine(kind, d, 1.0, ls, noise, c, Xall, Yall)
        sto = O.gpr_update(kind, 1.0, ls, noise, c, Xall, Yall)
        La, Wa, aa = eng.get_factor()
        Lf, Wf, af = full.get_factor()
        assert_close(La, sto.L, atol=floor, what=f"L after append k={k}")
        assert_close(La, Lf, rtol=1e-9, atol=floor, what="L append == full")
        ascale = max(1.0, np.abs(af).max())
        assert_close(aa, af, rtol=1e-7, atol=floor * ascale / min(noise, 1.0), what="alpha append == full")
        ma, va = eng.predict(Xq)
        mo, vo = O.predict(sto, Xq)
        assert_close(ma, mo, atol=floor, what="mean after append")
        assert_close(va, vo, atol=floor, what="var after append")
        assert_close(eng.eta(), O.eta_min_mean(sto), atol=floor, what="eta after append")
    # a hyper-parameter change invalidates the factor: append must refuse until set_data is called again
    eng.set_hyper(1.1, ls, noise, c)
    with pytest.raises(RuntimeError):
        eng.append_data(Xq[:2], np.zeros(2))
    with pytest.raises(ValueError):
        full.append_data(Xq[:2], np.zeros(3))


def test_destruction_order_and_sticky_errors_do_not_leak_into_later_calls():
    """A garbage collector may destroy a model handle before its trajectories (finalisers of a reference
    cycle run in arbitrary order): trajectory destruction must not touch the dead handle, and no cleanup
    error may surface from a later, unrelated launch check (hipGetLastError is sticky per host thread)."""
    _, obj, d, kind, N, noise = CONFIGS[0]
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=300)
    rng = np.random.default_rng(3)
    for _ in range(5):
        eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
        eng.use_private_stream()
        traj = eng.trajectory(rng.standard_normal((32, d)), rng.uniform(0, 6.28, 32), rng.standard_normal((32, 2)),
                              rng.standard_normal((N, 2)))
        assert traj(Xq).shape == (300, 2)
        eng.close()  # handle first ...
        with pytest.raises(RuntimeError):
            traj(Xq)
        traj.close()  # ... its trajectory afterwards
        other = _engine(kind, d, 1.0, ls, noise, c, X, Y)
        v, i = other.acq_topk("ei", other.eta(), Xq, 5)  # would report a stale "invalid device ordinal"
        assert len(i) == 5
        other.close()


@pytest.mark.parametrize("c", [c for c in CASES if c["noise"] >= 1e-3],
                         ids=[c["name"] for c in CASES if c["noise"] >= 1e-3])
def test_greedy_batch_pieces_match_mpmath_goldens(c):
    """tgp_set_penalization / tgp_penalization_values and tgp_clone_from + tgp_append_data (the engine's
    fantasized model) against the 50-digit vectors."""
    N, var0, noise = c["N"], c["variance"], c["noise"]
    floor = cancellation_floor(N + 3, var0, noise)
    X, Y = np.array(c["X"]), np.array(c["Y"])
    eng = _engine(c["kind"], c["d"], var0, c["lengthscales"], noise, c["mean_const"], X, Y)
    Xq, pend = np.array(c["Xq"]), np.array(c["Xg"])[1]
    r, sc = np.array(c["pen_radius"]), np.array(c["pen_scale"])
    for kind in ("soft", "hard"):
        with eng.penalized(kind, pend, r, sc):
            assert_close(eng.penalization_values(Xq), c["pen_" + kind], rtol=1e-12, atol=1e-300, what=kind)
            pei = eng.acq_values("ei", c["eta"], Xq)
        assert_close(pei, np.array(c["ei"]) * np.array(c["pen_" + kind]), atol=floor, what=f"{kind}-penalized ei")
    assert_close(eng.acq_values("ei", c["eta"], Xq), c["ei"], atol=floor, what="penalization cleared")
    twin = eng.clone()
    twin.append_data(pend, np.array(c["fant_y"]))
    fm, fv = twin.predict(Xq)
    assert_close(fm, c["fant_mean"], atol=floor / min(noise, 1.0) ** 0.5, what="fantasized mean")
    assert_close(fv, np.maximum(np.array(c["fant_var_raw"]), 1e-12), atol=floor, what="fantasized var")
    mean, var = eng.predict(Xq)  # the source is untouched
    assert_close(mean, c["mean"], atol=floor, what="base mean after clone")
    assert_close(var, c["var"], atol=floor, what="base var after clone")
    assert eng.N == N and twin.N == N + pend.shape[0]


@pytest.mark.parametrize("kind", ["soft", "hard"])
@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_penalized_sweeps_match_oracle(cfg, kind):
    """Local penalization on every sweep entry point: values, fused arg-max (routed through the value array),
    top-k, value-and-gradient (greedy_batch.py:250-389), with the LocalPenalization parameter recipe."""
    _, obj, d, kname, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kname, N, noise, M=3000)
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kname, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(11)
    pending = np.concatenate([rng.uniform(size=(4, d)), Xq[7:8]])  # one pending point IS a candidate
    lip, eta = O.lipschitz_estimate(st, np.concatenate([X, rng.uniform(size=(100, d))]))
    nm, ng = eng.acq_value_grad("nlcb", 0.0, np.concatenate([X, Xq[:50]]))  # the engine's own estimate inputs
    om, og = O.acq_value_and_grad(st, "nlcb", 0.0, np.concatenate([X, Xq[:50]]))
    assert_close(np.linalg.norm(ng, axis=1).max(), np.linalg.norm(og, axis=1).max(), rtol=1e-6, what="lipschitz")
    radius, scale = O.local_penalizer_parameters(st, pending, lip, eta)
    base = eng.acq_values("ei", eta, Xq)
    open_ = O.PENALIZERS[kind](Xq, pending, radius, scale)
    with eng.penalized(kind, pending, radius, scale):
        assert_close(eng.penalization_values(Xq), open_, rtol=1e-11, atol=1e-300, what="penalization")
        vals = eng.acq_values("ei", eta, Xq)
        val, idx, x = eng.acq_argmax("ei", eta, Xq, index_base=1000)
        tv, ti = eng.acq_topk("ei", eta, Xq, 9)
        gv, gg = eng.acq_value_grad("ei", eta, Xq[:64])
    assert_close(vals, base * open_, rtol=1e-11, atol=1e-300, what="penalized = base * phi")
    assert vals[7] <= base[7] * 0.5 + 1e-300  # at a pending point: Phi(-r/s) <= 1/2 (soft), 0 (hard)
    assert idx - 1000 == int(np.argmax(vals)) and val == vals[idx - 1000]
    np.testing.assert_array_equal(x, Xq[idx - 1000])
    ov_, oi_ = O.top_k(vals, 9)
    np.testing.assert_array_equal(ti, oi_)
    np.testing.assert_array_equal(tv, ov_)
    oval, ograd = O.penalized_value_and_grad(st, "ei", eta, kind, pending, radius, scale, Xq[:64])
    assert_close(gv, oval, atol=floor, what="penalized value")
    gscale = np.abs(ograd).max() + 1e-300
    assert_close(gg, ograd, rtol=1e-5, atol=max(floor * 1e3, 1e-9 * gscale), what="penalized gradient")
    assert np.all(np.isfinite(gg))
    # cleared on exit; an un-set penalization is an error for the stand-alone values
    assert_close(eng.acq_values("ei", eta, Xq), base, rtol=0, atol=0, what="cleared")
    with pytest.raises(RuntimeError):
        eng.penalization_values(Xq)
    with pytest.raises(ValueError):
        eng.set_penalization("soft", pending[:, :-1] if d > 1 else np.zeros((2, 3)), radius, scale)
    with pytest.raises(ValueError):
        eng.set_penalization("cubic", pending, radius, scale)


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_clone_then_append_is_the_fantasized_posterior(cfg):
    """tgp_clone_from: an independent copy (factor, data, hyper-parameters); conditioning the copy on pending
    points by tgp_append_data gives the posterior the reference's _fantasized_model computes through
    conditional_predict_f (greedy_batch.py:669-691, models.py:355-416); clone_from into an existing engine
    resets it; the source never changes."""
    _, obj, d, kind, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kind, N, noise, M=200)
    floor = cancellation_floor(N + 8, 1.0, noise)
    eng = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    m0, v0 = eng.predict(Xq)
    twin = eng.clone()
    np.testing.assert_array_equal(twin.predict(Xq)[0], m0)
    np.testing.assert_array_equal(twin.predict(Xq)[1], v0)
    for a, b in zip(twin.get_factor(), eng.get_factor()):
        np.testing.assert_array_equal(a, b)
    rng = np.random.default_rng(31)
    pend = rng.uniform(size=(6, d))
    kb = eng.predict_mean(pend)  # kriging believer
    twin.append_data(pend[:1], kb[:1])
    twin.append_data(pend[1:], kb[1:])  # the greedy loop's growth
    sto = O.fantasized_state(st, pend, O.predict(st, pend)[0])
    fm, fv = twin.predict(Xq)
    om, ov = O.predict(sto, Xq)
    assert_close(fm, om, atol=floor, what="fantasized mean")
    assert_close(fv, ov, atol=floor, what="fantasized var")
    if noise >= 1e-3:
        cm, cv = O.conditional_predict_f(st, Xq, pend, O.predict(st, pend)[0])
        assert_close(fm, cm, atol=floor, what="== conditional_predict_f mean")
        assert_close(fv, np.maximum(cv, 1e-12), atol=floor, what="== conditional_predict_f var")
    assert_close(fm, m0, atol=max(floor * 1e4, 1e-6), what="kriging believer keeps the mean")  # test_greedy_batch.py:233-257
    assert np.all(fv <= v0 + floor)  # :260-296
    assert_close(twin.eta(), O.eta_min_mean(sto), atol=floor, what="fantasized eta")
    np.testing.assert_array_equal(eng.predict(Xq)[0], m0)
    assert eng.N == N and twin.N == N + 6
    twin.clone_from(eng)  # reset
    assert twin.N == N
    np.testing.assert_array_equal(twin.predict(Xq)[1], v0)
    from trieste_amd.engine import GPEngine

    other = GPEngine(d + 1, kind)
    with pytest.raises(ValueError):
        other.clone_from(eng)
    fresh = GPEngine(d, kind)
    with pytest.raises(RuntimeError):
        eng.clone_from(fresh)  # the source has no hyper-parameters
    np.testing.assert_array_equal(eng.predict(Xq)[0], m0)


@pytest.mark.parametrize("c", [c for c in CASES if c["noise"] >= 1e-3],
                         ids=[c["name"] for c in CASES if c["noise"] >= 1e-3])
def test_entropy_tails_match_mpmath_goldens(c):
    """TGP_ACQ_MES / TGP_ACQ_GIBBON (+ tgp_set_repulsion against a clone conditioned on the pending points) vs the
    50-digit vectors, in the regime float64 resolves (gamma <= 30, see tests/test_oracle_golden.py)."""
    N, var0, noise = c["N"], c["variance"], c["noise"]
    floor = cancellation_floor(N + 3, var0, noise)
    X, Y = np.array(c["X"]), np.array(c["Y"])
    eng = _engine(c["kind"], c["d"], var0, c["lengthscales"], noise, c["mean_const"], X, Y)
    Xq, pend = np.array(c["Xq"]), np.array(c["Xg"])[1]
    smp = np.array(c["ent_samples"])
    gm, gv = np.array(c["mean"]), np.array(c["var"])
    gmax = np.max((smp[None, :] - gm[:, None]) / np.sqrt(gv)[:, None], axis=1)
    ok = gmax <= 30.0
    # sensitivity of the tails to the posterior's own rounding: d/dmean ~ gamma / sd
    sens = floor * 10 * (1.0 + np.abs(gmax[ok])) / np.sqrt(gv[ok])
    with pytest.raises(RuntimeError):
        eng.acq_values("mes", 0.0, Xq)  # no samples yet
    eng.set_min_value_samples(smp)
    assert_close(eng.acq_values("mes", 0.0, Xq)[ok], np.array(c["mes"])[ok], rtol=1e-6, atol=sens + 1e-14, what="mes")
    assert_close(eng.acq_values("gibbon", 0.0, Xq)[ok], np.array(c["gibbon_quality"])[ok], rtol=1e-6, atol=sens + 1e-14,
                 what="gibbon quality")
    twin = eng.clone()
    twin.append_data(pend, np.zeros(len(pend)))
    eng.set_repulsion(twin, 1.0 / len(pend) ** 2)
    both = eng.acq_values("gibbon", 0.0, Xq)
    eng.set_repulsion(None)
    quality = eng.acq_values("gibbon", 0.0, Xq)
    assert_close(both - quality, c["gibbon_repulsion"], atol=floor / noise + 1e-14, what="gibbon repulsion")
    eng.set_min_value_samples([])
    with pytest.raises(RuntimeError):
        eng.acq_values("gibbon", 0.0, Xq)


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_entropy_sweeps_and_gradients_match_oracle(cfg):
    """MES / GIBBON on every sweep entry point (values, arg-max, top-k, value-and-gradient), GIBBON with the
    repulsion twin, against the oracle's reference-form values (block determinant) and analytic gradients."""
    _, obj, d, kname, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kname, N, noise, M=2500)
    floor = cancellation_floor(N, 1.0, noise)
    eng = _engine(kname, d, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(13)
    eta = O.eta_min_mean(st)
    samples = eta - np.array([0.01, 0.05, 0.2, 0.35, 0.6])
    pending = rng.uniform(size=(4, d))
    weight = 1.0 / 16.0
    om, ov = O.predict(st, Xq)
    eng.set_min_value_samples(samples)
    twin = eng.clone()
    twin.append_data(pending, np.zeros(4))
    sto = O.fantasized_state(st, pending, np.zeros(4))
    gmax = np.max((samples[None, :] - om[:, None]) / np.sqrt(ov)[:, None], axis=1)
    ok = gmax <= 30.0
    sens = floor * 10 * (1.0 + np.abs(gmax)) / np.sqrt(ov) + 1e-13
    for acq, ref in (("mes", O.min_value_entropy_search(om, ov, samples)),
                     ("gibbon", O.gibbon_quality_term(om, ov, samples, noise))):
        vals = eng.acq_values(acq, 0.0, Xq)
        assert_close(vals[ok], ref[ok], rtol=1e-6, atol=sens[ok], what=acq)
        val, idx, x = eng.acq_argmax(acq, 0.0, Xq, index_base=7)
        assert idx - 7 == int(np.nanargmax(vals)) and val == vals[idx - 7]
        np.testing.assert_array_equal(x, Xq[idx - 7])
        tv, ti = eng.acq_topk(acq, 0.0, Xq, 11)
        ov_, oi_ = O.top_k(vals, 11)
        np.testing.assert_array_equal(ti, oi_)
        np.testing.assert_array_equal(tv, ov_)
    # batch GIBBON: quality + repulsion; the reference forms the repulsion from a block determinant
    eng.set_repulsion(twin, weight)
    full = eng.acq_values("gibbon", 0.0, Xq)
    ref_full = O.gibbon_quality_term(om, ov, samples, noise) + weight * 16.0 * O.gibbon_repulsion_term(st, Xq, pending, True)
    assert_close(full[ok], ref_full[ok], rtol=1e-6, atol=sens[ok] + floor / noise, what="gibbon + repulsion")
    # gradients at points near the best data (the regime L-BFGS-B works in), incl. one training input
    near = np.clip(X[np.argsort(Y)[:40]] + 0.03 * rng.standard_normal((40, d)), 0.0, 1.0)
    near[0] = X[np.argmin(Y)]
    for acq, tw, w in (("mes", None, 0.0), ("gibbon", None, 0.0), ("gibbon", sto, weight)):
        eng.set_repulsion(twin if tw is not None else None, w)
        gv_, gg = eng.acq_value_grad(acq, 0.0, near)
        oval, ograd = O.entropy_value_and_grad(st, acq, samples, near, tw, w)
        assert_close(gv_, oval, rtol=1e-6, atol=floor * 1e3 + 1e-12, what=f"{acq} value")
        gscale = np.abs(ograd).max() + 1e-300
        assert_close(gg, ograd, rtol=1e-5, atol=max(floor * 1e4, 1e-8 * gscale), what=f"{acq} gradient")
        sweep_vals = eng.acq_values(acq, 0.0, near)
        assert_close(gv_, sweep_vals, rtol=1e-8, atol=floor * 1e3 + 1e-12, what=f"{acq} value == sweep value")
    eng.set_repulsion(None)
    # local penalization composes with the entropy tails (LocalPenalization's second supported base)
    base = eng.acq_values("mes", 0.0, Xq)
    r, s = rng.uniform(0.1, 0.3, 4), rng.uniform(0.05, 0.2, 4)
    with eng.penalized("soft", pending, r, s):
        pen = eng.acq_values("mes", 0.0, Xq)
    assert_close(pen, base * O.soft_local_penalizer(Xq, pending, r, s), rtol=1e-11, atol=1e-300, what="penalized mes")
    # argument checks
    from trieste_amd.engine import GPEngine

    with pytest.raises(ValueError):
        eng.set_repulsion(GPEngine(d + 1, kname), 1.0)
    with pytest.raises(RuntimeError):
        eng.set_repulsion(GPEngine(d, kname), 1.0)  # a twin without data
    with pytest.raises(ValueError):
        eng.set_repulsion(twin, -1.0)


@pytest.mark.parametrize("N0", [1100, 2050])
def test_append_with_split_k_products_equals_full_refactorisation(N0):
    """Appending to a model with >= 1024 kept rows takes the split-k form of the three strip products
    (L21, Schur complement, T): same factor as a full refactorisation and as the oracle."""
    d, kind, noise = 5, "matern52", 1e-3
    X, Y = O.synthetic_problem(O.ackley, d, N0 + 40, seed=N0)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    eng = _engine(kind, d, 1.0, ls, noise, c, X[:N0], Y[:N0])
    eng.append_data(X[N0:N0 + 1], Y[N0:N0 + 1])
    eng.append_data(X[N0 + 1:], Y[N0 + 1:])
    full = _engine(kind, d, 1.0, ls, noise, c, X, Y)
    sto = O.gpr_update(kind, 1.0, ls, noise, c, X, Y)
    floor = cancellation_floor(N0 + 40, 1.0, noise)
    La, Wa, aa = eng.get_factor()
    Lf, Wf, af = full.get_factor()
    assert_close(La, sto.L, atol=floor, what="L after append")
    assert_close(La, Lf, rtol=1e-9, atol=floor, what="L append == full")
    assert_close(Wa, Wf, rtol=1e-7, atol=floor * 1e3 / noise ** 0.5, what="W append == full")
    assert_close(aa, af, rtol=1e-7, atol=floor * max(1.0, np.abs(af).max()) / noise, what="alpha append == full")
    Xq = np.random.default_rng(1).uniform(size=(300, d))
    ma, va = eng.predict(Xq)
    mo, vo = O.predict(sto, Xq)
    assert_close(ma, mo, atol=floor, what="mean after append")
    assert_close(va, vo, atol=floor, what="var after append")


@pytest.mark.parametrize("M,k", [(100, 100), (8000, 80), (8192, 1024), (8193, 160), (20000, 333), (65536, 1024),
                                 (70000, 50)])
def test_topk_by_sorting_equals_stable_descending_sort(M, k):
    """tgp_acq_topk: one sorting workgroup (M <= 8192), chunk sort + merge (M <= 65536), threshold passes above; all
    equal the stable descending sort of the engine's own values (value desc, index asc; optimizer.py:326-335),
    with duplicated candidates (ties) across chunk boundaries and an index base."""
    d, kind = 3, "matern52"
    X, Y = O.synthetic_problem(O.ackley, d, 90)
    eng = _engine(kind, d, 1.0, O.default_lengthscales(d), 1e-2, float(np.mean(Y)), X, Y)
    rng = np.random.default_rng(M)
    Xq = rng.uniform(size=(M, d))
    dup = rng.integers(0, M, size=min(M, 4000))
    Xq[dup] = Xq[(dup * 7 + 8191) % M]  # exact ties, many of them in different 8192-chunks
    eta = eng.eta()
    vals = eng.acq_values("ei", eta, Xq)
    tv, ti = eng.acq_topk("ei", eta, Xq, k, index_base=12345)
    ov, oi = O.top_k(vals, k)
    np.testing.assert_array_equal(ti, oi + 12345)
    np.testing.assert_array_equal(tv, ov)


@pytest.mark.parametrize("cfg", CONFIGS[:5], ids=[c[0] for c in CONFIGS[:5]])
def test_gibbon_repulsion_by_rank_m_update_equals_the_twin_sweep(cfg, monkeypatch):
    """A twin that is this model + m <= 16 appended rows gives its variance as var - sum_r (W'_r . k')^2 (m kernel
    sums per candidate); any other twin -- more rows, or forced with TGP_NO_LOWRANK -- is swept itself.  Both equal
    the oracle's block-determinant form; a changed base model or twin invalidates the cached check."""
    _, obj, d, kname, N, noise = cfg
    X, Y, ls, c, st, Xq = _problem(obj, d, kname, N, noise, M=1200)
    floor = cancellation_floor(N, 1.0, noise)
    rng = np.random.default_rng(17)
    samples = O.eta_min_mean(st) - np.array([0.02, 0.1, 0.3])
    om, ov = O.predict(st, Xq)
    gmax = np.max((samples[None, :] - om[:, None]) / np.sqrt(ov)[:, None], axis=1)
    ok = gmax <= 30.0

    def batch_values(pending, weight):
        eng = _engine(kname, d, 1.0, ls, noise, c, X, Y)
        eng.set_min_value_samples(samples)
        twin = eng.clone()
        twin.append_data(pending[:1], np.zeros(1))
        if len(pending) > 1:
            twin.append_data(pending[1:], rng.standard_normal(len(pending) - 1))  # observations do not matter
        eng.set_repulsion(twin, weight)
        return eng, twin, eng.acq_values("gibbon", 0.0, Xq)

    for m in (1, 5, 16, 17):  # 17 rows: beyond the rank-m form, the twin is swept
        pending = rng.uniform(size=(m, d))
        w = 1.0 / m ** 2
        ref = O.gibbon_quality_term(om, ov, samples, noise) + O.gibbon_repulsion_term(st, Xq, pending, True)
        eng, twin, low = batch_values(pending, w)
        monkeypatch.setenv("TGP_NO_LOWRANK", "1")
        _, _, swept = batch_values(pending, w)
        monkeypatch.delenv("TGP_NO_LOWRANK")
        sens = floor * 10 * (1.0 + np.abs(gmax)) / np.sqrt(ov) + floor / noise + 1e-13
        assert_close(low[ok], ref[ok], rtol=1e-6, atol=sens[ok], what=f"rank-{m} form == reference form")
        assert_close(low, swept, rtol=1e-7, atol=floor / noise + 1e-13, what=f"rank-{m} form == twin sweep")
        # the cached check follows the data: grow the twin, then change the base model
        extra = rng.uniform(size=(1, d))
        twin.append_data(extra, np.zeros(1))
        grown = eng.acq_values("gibbon", 0.0, Xq)
        ref2 = O.gibbon_quality_term(om, ov, samples, noise) + w * (m + 1) ** 2 * O.gibbon_repulsion_term(
            st, Xq, np.concatenate([pending, extra]), True)
        assert_close(grown[ok], ref2[ok], rtol=1e-6, atol=sens[ok], what="after growing the twin")
        eng.set_data(X[:-3], Y[:-3])  # the twin is no longer this model + rows: swept, whatever it is
        st3 = O.gpr_update(kname, 1.0, ls, noise, c, X[:-3], Y[:-3])
        m3, v3 = O.predict(st3, Xq)
        _, vt = O.predict(O.fantasized_state(st, np.concatenate([pending, extra]), np.zeros(m + 1)), Xq)
        ref3 = O.gibbon_quality_term(m3, v3, samples, noise) + 0.5 * w * (np.log(vt + noise) - np.log(v3 + noise))
        g3 = np.max((samples[None, :] - m3[:, None]) / np.sqrt(v3)[:, None], axis=1) <= 30.0
        assert_close(eng.acq_values("gibbon", 0.0, Xq)[g3], ref3[g3], rtol=1e-6, atol=sens[g3] * 10, what="foreign twin")


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [1, 15, 16, 17, 127, 128, 129, 255, 256, 257, 383, 384, 640])
def test_factor_at_the_leaf_and_panel_boundaries(N):
    """`update` around the sizes where the 128-leaf's structure changes (16-row panels, two-row-set panels, one vs
    several leaves, padding rows inside a leaf): L, W = L^-1 and alpha against the oracle."""
    rng = np.random.default_rng(100 + N)
    d, noise = 3, 1e-3
    X = rng.uniform(size=(N, d))
    Y = np.sin(3.0 * X.sum(axis=1)) + 0.1 * rng.standard_normal(N)
    ls = np.array([0.4, 0.5, 0.6])
    eng = _engine("matern52", d, 1.3, ls, noise, 0.2, X, Y)
    st = O.gpr_update("matern52", 1.3, ls, noise, 0.2, X, Y)
    L, W, alpha = eng.get_factor()
    floor = cancellation_floor(N, 1.3, noise)
    assert_close(L, st.L, atol=floor, what="L")
    assert np.array_equal(L, np.tril(L)) and np.array_equal(W, np.tril(W)), "zeros above the diagonals"
    assert_close(W @ st.L, np.eye(N), rtol=0, atol=1e-9 * (1 + 1.3 / noise), what="W L = I")
    import scipy.linalg as sla

    oalpha = sla.cho_solve((st.L, True), st.err)
    assert_close(alpha, oalpha, atol=floor * max(1.0, np.abs(oalpha).max()) / noise, what="alpha")
    # a second, different factorisation in the same buffers (L / W are wiped once per allocation, not per update)
    Y2 = np.cos(2.0 * X[:, 0]) - X[:, 1]
    eng.set_data(X[::-1].copy(), Y2[::-1].copy())
    st2 = O.gpr_update("matern52", 1.3, ls, noise, 0.2, X[::-1], Y2[::-1])
    L2, W2, _ = eng.get_factor()
    assert_close(L2, st2.L, atol=floor, what="L (second update)")
    assert np.array_equal(L2, np.tril(L2)) and np.array_equal(W2, np.tril(W2))
