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
    budget = i8x4_variance_bound(N, 1.0, np.abs(eng.get_factor()[1]).max())
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


@pytest.mark.parametrize("precision", ["i8x4", "i8x5"])
@pytest.mark.parametrize("noise", [1e-2, 1e-5])
def test_i8x4_at_n4096_against_the_float64_engine(noise, precision):
    """Full size: the digit-plane error against the float64 kernel on 20000 Philox candidates plus candidates at / next
    to training inputs; the float64 kernel itself is pinned to the CPU restatement at this size in test_gpu_c3.py."""
    import torch

    from trieste_amd.engine import GPEngine

    N, d = 4096, 8
    X, Y = O.synthetic_problem(O.ackley, d, N)
    ls = O.default_lengthscales(d)
    eng = GPEngine(d, "matern52")
    eng.set_hyper(1.0, ls, noise, float(np.mean(Y)))
    eng.set_data(X, Y)
    Xq = eng.sample_box(5678, 0, 20000, 0.0, 1.0)
    Xq[:64] = torch.from_numpy(X[:64]).cuda()
    Xq[64:128] = torch.from_numpy(X[64:128] + 1e-5).cuda()
    floor = cancellation_floor(N, 1.0, noise)
    fm, fv = eng.predict(Xq)
    eta = eng.eta()
    fei = eng.acq_values("ei", eta, Xq)
    eng.set_precision(precision)
    m, v = eng.predict(Xq)
    ei = eng.acq_values("ei", eta, Xq)
    assert_close(v.cpu().numpy(), fv.cpu().numpy(), atol=floor, what="var vs f64 engine")
    rel = float(np.max(np.abs(v.cpu().numpy() - fv.cpu().numpy()) / fv.cpu().numpy()))
    print(f"[{precision}] N=4096 noise={noise:g}: max pure relative |d var| = {rel:.3g}")
    if precision == "i8x5":
        assert rel <= 1e-5 or noise < 1e-4, rel   # the north-star's bar without any floor (borderline at 1e-5 noise)
    assert_close(ei.cpu().numpy(), fei.cpu().numpy(), atol=floor * 10, what="ei vs f64 engine")
    np.testing.assert_allclose(m.cpu().numpy(), fm.cpu().numpy(), rtol=1e-12, atol=1e-12)
    a = eng.acq_argmax("ei", eta, Xq)
    eng.set_precision("f64")
    b = eng.acq_argmax("ei", eta, Xq)
    assert a[1] == b[1] or abs(a[0] - b[0]) <= 1e-5 * abs(b[0]) + floor * 10
