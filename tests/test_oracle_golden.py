"""The numpy oracle vs the 50-digit mpmath golden vectors (runs on CPU, no GPU needed)."""
import numpy as np
import pytest

from oracle import gp_oracle as O
from tests.util import assert_close, cancellation_floor, load_goldens, load_wide_qei_goldens, reparam_sample_atol

CASES = load_goldens()


def _state(c):
    return O.gpr_update(c["kind"], c["variance"], np.array(c["lengthscales"]), c["noise"],
                        c["mean_const"], np.array(c["X"]), np.array(c["Y"]))


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_mpmath(c):
    st = _state(c)
    N, var0, noise = c["N"], c["variance"], c["noise"]
    floor = cancellation_floor(N, var0, noise)
    assert_close(st.L, np.array(c["L"]), atol=floor, what="L")
    Xq = np.array(c["Xq"])
    mean, var = O.predict(st, Xq)
    _, var_raw = O.predict(st, Xq, clip=False)
    assert_close(mean, c["mean"], atol=floor * 10, what="mean")
    assert_close(var_raw, c["var_raw"], atol=floor, what="var_raw")
    assert_close(var, c["var"], atol=floor, what="var")
    eta = O.eta_min_mean(st)
    assert_close(eta, c["eta"], atol=floor * 10, what="eta")
    # acquisition tails from the GOLDEN mean/var (isolates the tail arithmetic) ...
    gm, gv = np.array(c["mean"]), np.array(c["var"])
    assert_close(O.expected_improvement(gm, gv, c["eta"]), c["ei"], atol=1e-300, what="ei(golden mv)")
    assert_close(O.probability_of_improvement(gm, gv, c["eta"]), c["pi"], atol=1e-300, what="pi")
    assert_close(O.negative_lower_confidence_bound(gm, gv), c["nlcb"], what="nlcb")
    assert_close(O.augmented_expected_improvement(gm, gv, c["eta"], noise), c["aei"], atol=1e-300, what="aei")
    # cross-covariance block (models.py:188-254)
    n1, n2 = len(c["cov12"]), len(c["cov12"][0])
    assert_close(O.covariance_between_points(st, Xq[:n1], Xq[n1:n1 + n2]), c["cov12"], atol=floor, what="cov12")
    # joint
    jm, jc = O.predict_joint(st, np.array(c["Xg"]))
    assert_close(jm, c["joint_mean"], atol=floor * 10, what="joint mean")
    assert_close(jc, c["joint_cov"], atol=floor, what="joint cov")
    # trajectory
    W, b = np.array(c["rff_W"]), np.array(c["rff_b"])
    w, xi = np.array(c["traj_w"]), np.array(c["traj_xi"])
    v = O.decoupled_weights(st, W, b, w, xi)
    scale = max(1.0, np.max(np.abs(np.array(c["traj_v"]))))
    assert_close(v, c["traj_v"], atol=floor * scale / min(noise, 1.0) , what="traj v")
    tv = O.trajectory_eval(st, W, b, w, np.array(c["traj_v"]), Xq)
    assert_close(tv, c["traj"], atol=1e-9 * scale, what="trajectory")


@pytest.mark.parametrize("c", [c for c in CASES if c["noise"] >= 1e-3],
                         ids=[c["name"] for c in CASES if c["noise"] >= 1e-3])
def test_oracle_ei_and_qei_end_to_end(c):
    """Well-conditioned cases: full pipeline (own posterior -> EI / qEI) within 1e-5 relative."""
    st = _state(c)
    floor = cancellation_floor(c["N"], c["variance"], c["noise"])
    ei = O.ei_values(st, np.array(c["Xq"]), c["eta"])
    assert_close(ei, c["ei"], atol=floor, what="ei")
    qei = O.batch_mc_ei(st, np.array(c["Xg"]), np.array(c["eps"]), c["eta"], c["jitter"])
    assert_close(qei, c["qei"], atol=floor * 10, what="qei")


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_greedy_batch_pieces_match_mpmath(c):
    """Local penalizers (greedy_batch.py:341-354, 376-389) and the posterior a fantasized model represents
    (greedy_batch.py:630-773 through models.py:355-416) against the mpmath vectors; the reference's
    conditional formula and the refit on data + fantasized data must be the same posterior."""
    st = _state(c)
    floor = cancellation_floor(c["N"] + 3, c["variance"], c["noise"])
    Xq, pend = np.array(c["Xq"]), np.array(c["Xg"])[1]
    r, sc = np.array(c["pen_radius"]), np.array(c["pen_scale"])
    assert_close(O.soft_local_penalizer(Xq, pend, r, sc), c["pen_soft"], atol=1e-300, what="soft penalizer")
    assert_close(O.hard_local_penalizer(Xq, pend, r, sc), c["pen_hard"], atol=1e-300, what="hard penalizer")
    base = np.array(c["ei"])
    assert_close(O.penalized_acquisition(base, np.array(c["pen_soft"])), base * np.array(c["pen_soft"]), atol=1e-300,
                 what="exp(log a + log phi) = a phi")
    yf = np.array(c["fant_y"])
    fm, fv = O.predict(O.fantasized_state(st, pend, yf), Xq, clip=False)
    assert_close(fm, c["fant_mean"], atol=floor * 10 / min(c["noise"], 1.0) ** 0.5, what="fantasized mean")
    assert_close(fv, c["fant_var_raw"], atol=floor, what="fantasized var")
    if c["noise"] >= 1e-3:  # the reference's conditional form subtracts two nearly equal terms at tiny noise
        cm, cv = O.conditional_predict_f(st, Xq, pend, yf)
        assert_close(cm, c["fant_mean"], atol=floor * 100, what="conditional mean")
        assert_close(np.maximum(cv, 1e-12), np.maximum(np.array(c["fant_var_raw"]), 1e-12), atol=floor * 10,
                     what="conditional var")


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_entropy_tails_match_mpmath(c):
    """MES (entropy.py:195-214), GIBBON quality (:479-500) and repulsion (:580-619) against the mpmath vectors; the
    repulsion also through the identity the engine uses (conditioned variance + noise = the block determinant)."""
    st = _state(c)
    floor = cancellation_floor(c["N"] + 3, c["variance"], c["noise"])
    Xq, pend = np.array(c["Xq"]), np.array(c["Xg"])[1]
    gm, gv = np.array(c["mean"]), np.array(c["var"])
    smp = np.array(c["ent_samples"])
    # -gamma ratio / 2 - log Phi(-gamma) subtracts two terms of size gamma^2 / 2 and takes the ratio from
    # exp(difference of two such terms) -- in float64 in the reference too: where a sample lies far ABOVE a
    # candidate's mean (gamma >> 1; wild extrapolations of the tiny-noise cases) float64 has no digits left, so
    # the vectors pin the regime gamma <= 30
    gmax = np.max((smp[None, :] - gm[:, None]) / np.sqrt(gv)[:, None], axis=1)
    ok = gmax <= 30.0
    assert ok.sum() >= 3
    atol = 1e-16 + 2e-16 * np.maximum(gmax[ok], 1.0) ** 4
    assert_close(O.min_value_entropy_search(gm, gv, smp)[ok], np.array(c["mes"])[ok], rtol=1e-6, atol=atol,
                 what="mes(golden mv)")
    assert_close(O.gibbon_quality_term(gm, gv, smp, c["noise"])[ok], np.array(c["gibbon_quality"])[ok], rtol=1e-6,
                 atol=atol, what="gibbon quality(golden mv)")
    if c["noise"] >= 1e-3:
        rel = floor / c["noise"]  # log(var + noise): absolute error of var over its magnitude
        assert_close(O.gibbon_repulsion_term(st, Xq, pend, True), c["gibbon_repulsion"], atol=rel, what="repulsion")
        _, v0 = O.predict(st, Xq)
        _, vt = O.predict(O.fantasized_state(st, pend, np.zeros(len(pend))), Xq)
        twin_form = 0.5 * (np.log(vt + c["noise"]) - np.log(v0 + c["noise"])) / len(pend) ** 2
        assert_close(twin_form, c["gibbon_repulsion"], atol=rel, what="repulsion through the conditioned model")


WIDE = load_wide_qei_goldens()


@pytest.mark.parametrize("c", WIDE, ids=[c["name"] for c in WIDE])
def test_oracle_wide_qei_matches_mpmath(c):
    """Batch Monte-Carlo EI at q = 9, 17, 33, 50 (the group sizes the engine's tail is instantiated for and BASELINE
    config 4's q): joint mean / covariance, the reparametrised samples and qEI against 50-digit arithmetic, at an
    incumbent where every value is O(1) (reference function.py:1181-1186, models/gpflow/sampler.py:276-287)."""
    st = O.gpr_update(c["kind"], c["variance"], np.array(c["lengthscales"]), c["noise"], c["mean_const"],
                      np.array(c["X"]), np.array(c["Y"]))
    floor = cancellation_floor(c["N"], c["variance"], c["noise"])
    Xg, eps = np.array(c["Xg"]), np.array(c["eps"])
    jm, jc = O.predict_joint(st, Xg)
    assert_close(jm, c["joint_mean"], atol=floor * 10, what="wide joint mean")
    if "joint_cov" in c:
        assert_close(jc, c["joint_cov"], atol=floor, what="wide joint cov")
    atol = reparam_sample_atol(jc, floor, eps)
    assert_close(O.batch_reparam_samples(st, Xg, eps, c["jitter"]), c["samples"], atol=atol, what=f"samples q={c['q']}")
    want = np.array(c["qei"])
    assert np.all(want > 0.1)
    assert_close(O.batch_mc_ei(st, Xg, eps, c["eta"], c["jitter"]), want, atol=atol, what=f"qEI q={c['q']}")
