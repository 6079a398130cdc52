"""The split-precision sweep (tgp_set_precision TGP_PREC_I8X4: W K* on the int8 matrix cores, four digit planes per
operand) against the oracle and against the float64 engine.  It is an EMULATED-precision option: its error on the
variance is the parity tolerance of the float64 path (1e-5 relative + the cancellation floor of tests/util.py) plus
the written truncation budget ``i8x4_variance_bound`` (2^-32 of the operands' scales, grows with max |W| ~ the
conditioning).  On the headline-class problems (N >= 1000, d = 8, and the full N = 4096 size at both noise levels)
it stays INSIDE the plain parity tolerance, which the tests assert separately; on the small ill-conditioned ones
(N = 50, noise 1e-3: max |W| ~ 30) it needs the budget.  Candidates at and next to training inputs, both noise
levels and every kernel family are covered."""
import numpy as np
import pytest

from oracle import gp_oracle as O
from tests.util import assert_close, cancellation_floor, i8x4_variance_bound

pytestmark = pytest.mark.gpu

CONFIGS = [
    ("branin_m52_N50", O.branin, 2, "matern52", 50, 1e-3),
    ("hartmann_rbf_N300", O.hartmann_6, 6, "rbf", 300, 1e-2),
    ("ackley8_m52_N1000", O.ackley, 8, "matern52", 1000, 1e-2),
    ("ackley8_m52_N1000_lownoise", O.ackley, 8, "matern52", 1000, 1e-5),
    ("ackley16_m32_N257", O.ackley, 16, "matern32", 257, 1e-3),
    ("ackley3_m12_N130", O.ackley, 3, "matern12", 130, 1e-3),
    ("ackley32_rbf_N600", O.ackley, 32, "rbf", 600, 1e-2),
]


def _setup(obj, d, kind, N, noise, M=1500):
    from trieste_amd.engine import GPEngine

    X, Y = O.synthetic_problem(obj, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    st = O.gpr_update(kind, 1.0, ls, noise, c, X, Y)
    rng = np.random.default_rng(5678)
    Xq = rng.uniform(size=(M, d))
    Xq[:5] = X[:5]
    Xq[5:10] = X[5:10] + 1e-6
    Xq[-3:] = 4.0 + rng.uniform(size=(3, d))
    eng = GPEngine(d, kind)
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    return eng, st, Xq


@pytest.mark.parametrize("precision", ["i8x4", "i8x5"])
@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_i8x4_sweep_is_inside_the_parity_tolerance(cfg, precision):
    _, obj, d, kind, N, noise = cfg
    eng, st, Xq = _setup(obj, d, kind, N, noise)
    floor = cancellation_floor(N, 1.0, noise)
    budget = i8x4_variance_bound(N, 1.0, np.abs(eng.get_factor()[1]).max())  # (tight scales since round 4: a quarter of r03's)
    if precision == "i8x5":
        if d > 16:
            with pytest.raises(ValueError):
                eng.set_precision("i8x5")   # five planes of a 64-candidate tile + dp = 32 coordinates exceed the LDS
            return
        budget = 0.0  # five planes (truncation at 2^-40 of the scales): the PLAIN parity tolerance, everywhere
    om, ov = O.predict(st, Xq)
    fm, fv = eng.predict(Xq)
    eng.set_precision(precision)
    mean, var = eng.predict(Xq)
    assert_close(mean, om, atol=floor * 10, what="mean")
    assert_close(var, ov, atol=floor + budget, what="var")
    worst = float(np.max(np.abs(np.asarray(var) - ov) / (1e-5 * np.abs(ov) + floor)))
    print(f"[{precision}] {cfg[0]}: max |d var| / parity tolerance = {worst:.4f}, budget / floor = {budget / floor:.2f}")
    if (N >= 1000 and d == 8) or precision == "i8x5":  # headline-class / five planes: inside the plain parity tolerance
        assert worst <= 1.0, worst
    np.testing.assert_allclose(mean, fm, rtol=1e-12, atol=1e-12)       # the mean never leaves float64
    eta = eng.eta()
    ei = eng.acq_values("ei", eta, Xq)
    oei = O.expected_improvement(om, ov, eta)
    assert_close(ei, oei, atol=floor * 10 + budget, what="ei")
    val, idx, x = eng.acq_argmax("ei", eta, Xq)
    assert idx == int(np.argmax(ei)) and val == ei[idx]
    oi = int(np.argmax(oei))
    assert idx == oi or abs(oei[oi] - oei[idx]) <= 1e-5 * oei[oi] + floor * 10 + budget
    tv, ti = eng.acq_topk("ei", eta, Xq, 9)
    ov_, oi_ = O.top_k(np.asarray(ei), 9)
    np.testing.assert_array_equal(ti, oi_)
    # ragged tails and tiny launches
    for m in (1, 63, 64, 65, 129):
        mm, vv = eng.predict(Xq[:m])
        np.testing.assert_array_equal(mm, mean[:m])
        np.testing.assert_array_equal(vv, var[:m])
    eng.set_precision("f64")
    m2, v2 = eng.predict(Xq)
    np.testing.assert_array_equal(v2, fv)                               # switching back restores the parity path


def _n4096(noise):
    """The headline-size model, 20000 Philox candidates + candidates at / next to training inputs, and the ORACLE's
    (mean, var, EI) on them: the reference-shaped CPU restatement (oracle/cpu_baseline.py TorchCpuSweep, validated
    against oracle/gp_oracle.py in tests/test_oracle_golden.py), not the float64 engine."""
    import torch

    from oracle.cpu_baseline import TorchCpuSweep
    from trieste_amd.engine import GPEngine

    N, d = 4096, 8
    X, Y = O.synthetic_problem(O.ackley, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    eng = GPEngine(d, "matern52")
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    Xq = eng.sample_box(5678, 0, 20000, 0.0, 1.0)
    Xq[:64] = torch.from_numpy(X[:64]).cuda()
    Xq[64:128] = torch.from_numpy(X[64:128] + 1e-5).cuda()
    st = O.gpr_update("matern52", 1.0, ls, noise, c, X, Y)
    sw = TorchCpuSweep(st)
    host = Xq.cpu().numpy()
    mv = [sw.chunk_mean_var(host[s:s + 10000]) for s in range(0, host.shape[0], 10000)]
    om = np.concatenate([m.numpy() for m, _ in mv])
    ov = np.concatenate([v.numpy() for _, v in mv])
    eta = O.eta_min_mean(st)
    oei = np.concatenate([sw.chunk_values(host[s:s + 10000], eta, improved=False).numpy()
                          for s in range(0, host.shape[0], 10000)])
    return eng, Xq, om, ov, oei, eta, cancellation_floor(N, 1.0, noise)


@pytest.mark.parametrize("noise", [1e-2, 1e-5])
def test_split_precision_at_n4096_against_the_oracle(noise):
    """Full size, ALL THREE arithmetics against the CPU restatement under the PLAIN parity tolerance (no extra budget):
    variance, mean, EI and the arg-max.  The observed worst error / tolerance is printed per arithmetic."""
    eng, Xq, om, ov, oei, eta, floor = _n4096(noise)
    oi = int(np.argmax(oei))
    for precision in ("f64", "i8x4", "i8x5"):
        eng.set_precision(precision)
        m, v = (t.cpu().numpy() for t in eng.predict(Xq))
        ei = eng.acq_values("ei", eta, Xq).cpu().numpy()
        rv = float(np.max(np.abs(v - ov) / (1e-5 * np.abs(ov) + floor)))
        rm = float(np.max(np.abs(m - om) / (1e-5 * np.abs(om) + floor * 10)))
        re_ = float(np.max(np.abs(ei - oei) / (1e-5 * np.abs(oei) + floor)))
        print(f"[margin] n4096 noise={noise:g} {precision}: var {rv:.3g}  mean {rm:.3g}  ei {re_:.3g}  (x tolerance)")
        assert_close(v, ov, atol=floor, what=f"{precision} var vs oracle")
        assert_close(m, om, atol=floor * 10, what=f"{precision} mean vs oracle")
        assert_close(ei, oei, atol=floor, what=f"{precision} ei vs oracle")
        val, idx, _ = eng.acq_argmax("ei", eta, Xq)
        assert idx == oi or abs(oei[oi] - oei[idx]) <= 1e-5 * oei[oi] + floor, (precision, idx, oi)
    eng.set_precision("f64")


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_auto_precision_stays_inside_the_plain_tolerance(cfg):
    """TGP_PREC_AUTO (round 4): the int8 sweep with the a-posteriori repair.  On every parity configuration -- also the
    small ill-conditioned ones the plain four-plane sweep fails -- mean, variance and EI hold the PLAIN parity tolerance
    against the oracle candidate by candidate, on EVERY rung the ladder visits, and the fused arg-max returns the float64
    sweep's winner (its index; its value from the float64 kernel, to the summation order of the row-group split)."""
    _, obj, d, kind, N, noise = cfg
    eng, st, Xq = _setup(obj, d, kind, N, noise)
    floor = cancellation_floor(N, 1.0, noise)
    om, ov = O.predict(st, Xq)
    eta = eng.eta()
    oei = O.expected_improvement(om, ov, eta)
    eta_mid = float(np.median(om))           # a second incumbent with O(1) values everywhere
    oei_mid = O.expected_improvement(om, ov, eta_mid)
    f64 = {e: eng.acq_argmax("ei", e, Xq)[:2] for e in (eta, eta_mid)}
    f64_topk = eng.acq_topk("ei", eta_mid, Xq, 5)
    eng.set_precision("auto")
    assert eng.get_precision()[:2] == ("auto", "i8x4")
    rungs = []
    for sweep in range(4):   # enough sweeps for the ladder to settle (at most two demotions)
        req, eff, _ = eng.get_precision()
        mean, var = eng.predict(Xq)
        frac = eng.get_precision()[2]
        rungs.append((eff, round(frac, 4)))
        worst = float(np.max(np.abs(np.asarray(var) - ov) / (1e-5 * np.abs(ov) + floor)))
        assert_close(var, ov, atol=floor, what=f"var under auto ({eff})")
        assert_close(mean, om, atol=floor * 10, what=f"mean under auto ({eff})")
        assert_close(eng.acq_values("ei", eta, Xq), oei, atol=floor, what=f"ei under auto ({eff})")
        assert_close(eng.acq_values("ei", eta_mid, Xq), oei_mid, atol=floor, what=f"ei (median eta) under auto ({eff})")
        for e in (eta, eta_mid):
            val, idx, _ = eng.acq_argmax("ei", e, Xq)
            # the float64 winner: its index, and its value as the float64 kernel computes it (the recomputation runs the
            # row-group-split instantiation: the same products, the partial sums of |c|^2 added in another order: 1e-13)
            assert idx == f64[e][1] and abs(val - f64[e][0]) <= 1e-12 * abs(f64[e][0]), (eff, e, val, idx, f64[e])
        tv, ti = eng.acq_topk("ei", eta_mid, Xq, 5)
        assert_close(tv, f64_topk[0], atol=floor, what="top-k values under auto")
        for m in (1, 63, 64, 65, 129):       # ragged tails and tiny launches
            mm, vv = eng.predict(Xq[:m])
            assert_close(vv, ov[:m], atol=floor, what=f"var under auto, M={m}")
    print(f"[margin] auto {cfg[0]}: rungs (arithmetic, recomputed fraction) {rungs}; last var error / tolerance {worst:.4f}")
    # the ladder only ever moves down (a move follows a sweep that recomputed more than 10 % of its candidates -- any of the
    # sweeps of an iteration, the arg-max ones recompute their band as well, so the recorded predict fraction is a hint)
    order = {"i8x4": 0, "i8x5": 1, "f64": 2}
    for (a, fa), (b, _) in zip(rungs, rungs[1:]):
        assert order[b] >= order[a], rungs
    if d > 16:
        assert all(r[0] != "i8x5" for r in rungs)
    # new hyper-parameters restart the ladder at four planes
    eng.set_hyper(1.0, O.default_lengthscales(d), 0.5, float(st.mean_const))
    X, Y = O.synthetic_problem(obj, d, N)
    eng.set_data(X, Y)
    assert eng.get_precision()[1] == "i8x4"
    st2 = O.gpr_update(kind, 1.0, O.default_lengthscales(d), 0.5, float(st.mean_const), X, Y)
    om2, ov2 = O.predict(st2, Xq)
    mean, var = eng.predict(Xq)
    assert_close(var, ov2, atol=cancellation_floor(N, 1.0, 0.5), what="var under auto after set_hyper")
    eng.set_precision("f64")
    assert eng.get_precision() == ("f64", "f64", -1.0)


def test_auto_precision_on_the_headline_model():
    """N = 4096, d = 8, Matern-5/2, both noise levels: AUTO stays on FOUR planes (no candidate of a 2^17 sweep needs the
    float64 recomputation for its tolerance; the arg-max band holds a handful), the arg-max is the float64 sweep's, and
    variance / EI hold the plain tolerance against the oracle (test_split_precision_at_n4096... covers i8x4
    without the repair)."""
    for noise in (1e-2, 1e-5):
        eng, Xq, om, ov, oei, eta, floor = _n4096(noise)
        want = eng.acq_argmax("ei", eta, Xq)[:2]
        eta_mid = float(np.median(om))
        want_mid = eng.acq_argmax("ei", eta_mid, Xq)[:2]
        eng.set_precision("auto")
        m, v = (t.cpu().numpy() for t in eng.predict(Xq))
        f_pred = eng.get_precision()[2]
        assert_close(v, ov, atol=floor, what="auto var vs oracle")
        assert_close(m, om, atol=floor * 10, what="auto mean vs oracle")
        assert_close(eng.acq_values("ei", eta, Xq).cpu().numpy(), oei, atol=floor, what="auto ei vs oracle")
        got = eng.acq_argmax("ei", eta, Xq)[:2]
        assert got[1] == want[1] and abs(got[0] - want[0]) <= 1e-12 * abs(want[0]), (got, want)
        f_arg = eng.get_precision()[2]
        got = eng.acq_argmax("ei", eta_mid, Xq)[:2]
        assert got[1] == want_mid[1] and abs(got[0] - want_mid[0]) <= 1e-12 * abs(want_mid[0]), (got, want_mid)
        req, eff, f_mid = eng.get_precision()
        print(f"[margin] auto on the headline model, noise {noise:g}: {eff}; recomputed fraction predict {f_pred:.2e}, "
              f"arg-max {f_arg:.2e}, arg-max at the median incumbent {f_mid:.2e}")
        assert eff == "i8x4" and max(f_pred, f_arg, f_mid) < 0.01
        # the canary: every sweep recomputed its sample (one candidate in 4096) in float64 and compared it with the bound the
        # int8 kernel priced it at -- nothing outside, the worst sample well inside, the ladder still on its first rung
        rep = eng.get_auto_report()
        print(f"[margin] auto canary on the headline model, noise {noise:g}: {rep}")
        assert rep["checked"] >= 4 * (Xq.shape[0] // 4096) and rep["violations"] == 0 and rep["demotions"] == 0, rep
        assert rep["level"] == 0 and 0.0 <= rep["worst_ratio"] < 1.0, rep
        eng.set_precision("f64")
        assert eng.get_auto_report()["level"] == -1


@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[0], CONFIGS[6]], ids=[CONFIGS[2][0], CONFIGS[0][0], CONFIGS[6][0]])
def test_auto_repair_paths_agree(cfg):
    """TGP_PREC_AUTO recomputes a SHORT list of candidates (up to 512: the canary's sample and a handful of flagged ones) as a
    product -- K*^T, W K*^T, column sums, tail -- and a longer one through the row-group-split sweep (variant bit 6 forces the
    sweep for every list).  Both are the float64 posterior of the same candidates: the values agree to the rounding of the
    two summation orders, the lists are the same, the winner is the same.  (The well-conditioned configurations recompute a
    candidate or two; the ill-conditioned one about half of its candidates in its first sweep -- ~300 of 600 through the
    product, ~800 of 1500 and ~3000 of 6000 through the sweep: both sides of the routing are met with lists of some length.)"""
    _, obj, d, kind, N, noise = cfg
    seen = []
    for M in (600, 1500, 6000):
        out = {}
        for variant in (0, 64):
            eng, st, Xq = _setup(obj, d, kind, N, noise, M=M)
            eta = eng.eta()
            eng.set_variant(variant)
            eng.set_precision("auto")
            mean, var = eng.predict(Xq)          # the FIRST sweep on four planes, whatever the ladder does afterwards
            eff, frac = eng.get_precision()[1:]
            val, idx, _ = eng.acq_argmax("ei", eta, Xq)
            out[variant] = (np.asarray(mean), np.asarray(var), frac, val, idx, eng.get_precision()[1])
            eng.set_precision("f64")
        a, b = out[0], out[64]
        assert a[2] == b[2] and a[5] == b[5], (a[2], b[2], a[5], b[5])   # the same list, the same rung afterwards
        np.testing.assert_allclose(a[0], b[0], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-11, atol=1e-13)   # var = sigma^2 - |c|^2: the cancellation's rounding
        assert a[4] == b[4] and abs(a[3] - b[3]) <= 1e-11 * abs(b[3]) + 1e-300, (a[3:5], b[3:5])
        seen.append(round(a[2] * M))
    print(f"[margin] auto repair paths, {cfg[0]}: recomputed candidates of the first sweeps {seen} (product path up to 512)")


@pytest.mark.parametrize("cfg", [CONFIGS[2], CONFIGS[3], CONFIGS[0]], ids=[CONFIGS[2][0], CONFIGS[3][0], CONFIGS[0][0]])
def test_auto_canary_fires_when_its_error_model_is_wrong_for_the_input(cfg):
    """The per-candidate bound of TGP_PREC_AUTO is K_SIGMA = 8 standard deviations of a STATISTICAL model of the dropped
    digit pairs.  K* is generated inside the kernel, so digits aligned on purpose cannot be fed through the boundary; what
    the run-time check must catch is `the float64 value lies outside the bound`, and that is driven here from the other
    side: with K_SIGMA = 0.02 the bound is 400 times tighter than the arithmetic's real error, i.e. the model is wrong for
    every candidate.  The canary (one candidate in 4096, recomputed in float64 inside the sweep and compared on the device)
    must fire on the first sweep, the ladder must leave its rungs -- five planes fail the same way -- and the values the
    SAME call returns must hold the plain parity tolerance: the synchronising entry points repeat their sweeps on the next
    rung before they return."""
    _, obj, d, kind, N, noise = cfg
    eng, st, Xq = _setup(obj, d, kind, N, noise, M=3 * 4096 + 77)
    floor = cancellation_floor(N, 1.0, noise)
    om, ov = O.predict(st, Xq)
    eta_mid = float(np.median(om))
    want = eng.acq_argmax("ei", eta_mid, Xq)[:2]
    eng.set_precision("auto")
    eng.set_auto_sigma(0.02)
    assert eng.get_precision()[:2] == ("auto", "i8x4") and eng.get_auto_report()["checked"] == 0
    mean, var = eng.predict(Xq)              # first sweep: samples break their bounds -> five planes -> float64 -> returned
    rep = eng.get_auto_report()
    print(f"[margin] auto canary, bound 400 x too tight, {cfg[0]}: {rep}; in effect {eng.get_precision()[1]}")
    assert rep["violations"] >= 1 and rep["demotions"] >= 1 and rep["worst_ratio"] > 1.0, rep
    assert rep["level"] >= 1 and eng.get_precision()[1] in ("i8x5", "f64"), rep
    assert_close(var, ov, atol=floor, what="var returned by the call whose canary fired")
    assert_close(mean, om, atol=floor * 10, what="mean returned by the call whose canary fired")
    got = eng.acq_argmax("ei", eta_mid, Xq)[:2]
    assert got[1] == want[1] and abs(got[0] - want[0]) <= 1e-12 * abs(want[0]), (got, want)
    # the fused arg-max repeats its sweep as well: a fresh ladder, first call
    eng.set_auto_sigma(0.02)
    assert eng.get_precision()[1] == "i8x4"
    got = eng.acq_argmax("ei", eta_mid, Xq)[:2]
    assert got[1] == want[1] and abs(got[0] - want[0]) <= 1e-12 * abs(want[0]), (got, want)
    assert eng.get_auto_report()["demotions"] >= 1
    # the default bound restarts the ladder and holds on the same input
    eng.set_auto_sigma(8.0)
    assert eng.get_precision()[1] == "i8x4"
    mean, var = eng.predict(Xq)
    assert_close(var, ov, atol=floor, what="var under auto, default K_SIGMA")
    rep = eng.get_auto_report()
    assert rep["violations"] == 0 and rep["demotions"] == 0 and rep["checked"] >= 3, rep
    with pytest.raises(ValueError):
        eng.set_auto_sigma(0.0)
    eng.set_precision("f64")


def test_auto_outputs_are_a_pure_function_of_model_and_inputs():
    """Round 6: the canary's samples are compared on the device and NOT written over the int8 results, and
    the uniform sample's offset is a function of (N, hyper-parameters, M, rung) instead of a call counter.  Hence two identical
    calls return identical bits, the ladder takes the same decisions, and a candidate's value does not depend on which other
    candidates share its call (only what a sweep FLAGS is replaced by its float64 value, and a flag is the candidate's own)."""
    _, obj, d, kind, N, noise = CONFIGS[2]
    eng, st, Xq = _setup(obj, d, kind, N, noise, M=3 * 4096 + 77)
    eta = float(np.median(O.predict(st, Xq)[0]))
    eng.set_precision("auto")
    runs = []
    for _ in range(3):
        m, v = (np.asarray(t).copy() for t in eng.predict(Xq))
        ei = np.asarray(eng.acq_values("ei", eta, Xq)).copy()
        runs.append((m, v, ei, eng.acq_argmax("ei", eta, Xq)[:2], eng.get_precision()[1]))
    for m, v, ei, win, eff in runs[1:]:
        assert eff == runs[0][4] == "i8x4"
        assert np.array_equal(m, runs[0][0]) and np.array_equal(v, runs[0][1]) and np.array_equal(ei, runs[0][2])
        assert win == runs[0][3], (win, runs[0][3])
    # a sub-batch (another M: other uniform samples, other adversarial picks): the shared candidates' values are the same bits
    for m_sub in (4096 + 100, 700, 64):
        ms, vs = (np.asarray(t) for t in eng.predict(Xq[:m_sub]))
        es = np.asarray(eng.acq_values("ei", eta, Xq[:m_sub]))
        assert np.array_equal(ms, runs[0][0][:m_sub]) and np.array_equal(vs, runs[0][1][:m_sub])
        assert np.array_equal(es, runs[0][2][:m_sub])
    # ... and a second engine on the same data returns them too (the offset does not depend on a handle's history)
    eng2, _, _ = _setup(obj, d, kind, N, noise, M=16)
    eng2.set_precision("auto")
    m2, v2 = (np.asarray(t) for t in eng2.predict(Xq))
    assert np.array_equal(m2, runs[0][0]) and np.array_equal(v2, runs[0][1])
    eng.set_precision("f64")
    eng2.set_precision("f64")


def test_auto_canary_strata_are_reported_apart():
    """Round 6: besides the uniform 1-in-4096 sample every AUTO sweep recomputes an ADVERSARIAL stratum -- of every 1 / 64 of
    the sweep the unflagged candidate whose bound sits closest to its tolerance (where a failure of the independence model
    would show first; the uniform sample meets those at the same 1 / 4096 as the far field) -- and compares it with its own
    bound on the device.  tgp_get_auto_strata reports the two apart; tgp_get_auto_report their sum."""
    _, obj, d, kind, N, noise = CONFIGS[2]
    M = 3 * 4096 + 77
    eng, st, Xq = _setup(obj, d, kind, N, noise, M=M)
    om, ov = O.predict(st, Xq)
    floor = cancellation_floor(N, 1.0, noise)
    eng.set_precision("auto")
    assert eng.get_auto_strata() == dict(uniform=dict(checked=0, violations=0, worst_ratio=0.0),
                                         adversarial=dict(checked=0, violations=0, worst_ratio=0.0), slack_saved=0)
    n_sweeps = 3
    for _ in range(n_sweeps):
        mean, var = eng.predict(Xq)
    s = eng.get_auto_strata()
    rep = eng.get_auto_report()
    print(f"[margin] auto canary strata, {CONFIGS[2][0]}: {s}")
    assert s["uniform"]["checked"] in (n_sweeps * (M // 4096), n_sweeps * (M // 4096 + 1)), s
    assert s["adversarial"]["checked"] == n_sweeps * 64, s       # 193 blocks of 64 candidates: every group has a pick
    assert s["uniform"]["violations"] == s["adversarial"]["violations"] == 0 and rep["level"] == 0, (s, rep)
    assert rep["checked"] == s["uniform"]["checked"] + s["adversarial"]["checked"]
    assert 0.0 < s["adversarial"]["worst_ratio"] < 1.0 and 0.0 <= s["uniform"]["worst_ratio"] < 1.0, s
    assert_close(var, ov, atol=floor, what="var under auto with both strata")
    # a tiny sweep: one block -> one adversarial pick, no uniform sample unless the offset hits
    eng.predict(Xq[:40])
    s2 = eng.get_auto_strata()
    assert s2["adversarial"]["checked"] == s["adversarial"]["checked"] + 1, (s, s2)
    # the bound made 400 x too tight: the adversarial stratum fires as well (it samples where the bound / tolerance is LARGEST,
    # the violation ratio is |d var| / bound -- any candidate breaks a bound that tight)
    eng.set_auto_sigma(0.02)
    assert eng.get_auto_strata()["adversarial"]["checked"] == 0      # tgp_set_auto_sigma clears the report
    mean, var = eng.predict(Xq)
    s3 = eng.get_auto_strata()
    print(f"[margin] auto canary strata, bound 400 x too tight: {s3}; in effect {eng.get_precision()[1]}")
    assert s3["adversarial"]["violations"] >= 1 and s3["uniform"]["violations"] >= 1, s3
    assert eng.get_auto_report()["demotions"] >= 1
    assert_close(var, ov, atol=floor, what="var returned by the call whose canaries fired")
    eng.set_precision("f64")


def test_auto_ladder_survives_the_trial_evaluations_of_a_refit():
    """tgp_set_hyper used to decide keep-or-restart on the spot, so a fit's trial evaluations (prior draws, L-BFGS-B
    steps: far-away hyper-parameters, then back) restarted a ladder whose rung the refit itself would have kept -- and the
    failed rung plus its repeat sweep were paid again at every BO step.  The decision is now taken at the next SWEEP from the
    hyper-parameters in effect then; what the closed epoch's canary counted joins the report's totals."""
    _, obj, d, kind, N, noise = CONFIGS[0]        # ill-conditioned: four planes recompute ~half of the candidates -> five planes
    eng, st, Xq = _setup(obj, d, kind, N, noise, M=1500)
    X, Y = O.synthetic_problem(obj, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    eng.set_precision("auto")
    for _ in range(3):
        eng.predict(Xq)
    level = eng.get_auto_report()["level"]
    checked = eng.get_auto_report()["checked"]
    assert level >= 1 and checked > 0, eng.get_auto_report()
    # a "fit": far-away trial hyper-parameters, then values within a factor two of where the rung was left
    for trial in (dict(v=30.0, l=0.05, n=1e-6), dict(v=0.01, l=7.0, n=0.5)):
        eng.set_hyper(trial["v"], ls * trial["l"], trial["n"], c)
        eng.nlml_trial()
    eng.set_hyper(1.0, ls * 1.2, noise * 1.3, c)
    eng.set_data(X, Y)
    assert eng.get_auto_report()["level"] == level, eng.get_auto_report()      # kept: no sweep ran under the trial values
    assert eng.get_auto_report()["checked"] >= checked                          # the closed epochs' samples are in the totals
    st2 = O.gpr_update(kind, 1.0, ls * 1.2, noise * 1.3, c, X, Y)
    mean, var = eng.predict(Xq)
    assert_close(var, O.predict(st2, Xq)[1], atol=cancellation_floor(N, 1.0, noise * 1.3), what="var after the kept refit")
    assert eng.get_auto_report()["checked"] > checked
    # hyper-parameters far from where the rung was left restart the ladder at four planes -- at the next sweep
    eng.set_hyper(1.0, ls, 0.5, c)
    eng.set_data(X, Y)
    assert eng.get_precision()[1] == "i8x4"
    # a clone takes the source's rung along
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    for _ in range(3):
        eng.predict(Xq)
    level = eng.get_auto_report()["level"]
    assert level >= 1
    from trieste_amd.engine import GPEngine
    twin = GPEngine(d, kind)
    twin.set_precision("auto")
    twin.clone_from(eng)
    assert twin.get_auto_report()["level"] == level
    eng.set_precision("f64")
