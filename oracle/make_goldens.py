"""Generate 50-digit mpmath golden vectors for the GP hot path -> tests/golden/gp_goldens.json.

TEST INFRASTRUCTURE.  Run from the repo root:  python oracle/make_goldens.py

Why mpmath and not the reference: trieste's arithmetic for this path lives in GPflow /
GPflux / TensorFlow(-Probability), none of which is installed (or installable) here, and the
reference's own tests hold no numeric golden vectors.  The goldens are
therefore computed from the published formulas (with the reference call
sites cited in oracle/gp_oracle.py) in 50-digit arithmetic, rounded once to float64.

Every quantity is computed independently of oracle/gp_oracle.py (own Cholesky, own
substitutions, mpmath ncdf/npdf), so the goldens pin the numpy oracle as well as the HIP engine.
"""
from __future__ import annotations

import json
import os
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 50


def mpf(x):
    return mp.mpf(float(x))  # inputs are float64 values, taken exactly


def kern(kind, variance, ls, a, b):
    r2 = mp.mpf(0)
    for k in range(len(a)):
        t = (mpf(a[k]) - mpf(b[k])) / mpf(ls[k])
        r2 += t * t
    v = mpf(variance)
    if kind == "rbf":
        return v * mp.exp(-r2 / 2)
    r = mp.sqrt(r2)
    if kind == "matern12":
        return v * mp.exp(-r)
    if kind == "matern32":
        s = mp.sqrt(3)
        return v * (1 + s * r) * mp.exp(-s * r)
    if kind == "matern52":
        s = mp.sqrt(5)
        return v * (1 + s * r + mp.mpf(5) / 3 * r2) * mp.exp(-s * r)
    raise ValueError(kind)


def chol(A):
    n = A.rows
    L = mp.zeros(n, n)
    for j in range(n):
        s = A[j, j] - sum(L[j, k] ** 2 for k in range(j))
        L[j, j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - sum(L[i, k] * L[j, k] for k in range(j))) / L[j, j]
    return L


def fsub(L, b):  # L y = b
    n = L.rows
    y = [mp.mpf(0)] * n
    for i in range(n):
        y[i] = (b[i] - sum(L[i, k] * y[k] for k in range(i))) / L[i, i]
    return y


def bsub(L, y):  # L^T x = y
    n = L.rows
    x = [mp.mpf(0)] * n
    for i in reversed(range(n)):
        x[i] = (y[i] - sum(L[k, i] * x[k] for k in range(i + 1, n))) / L[i, i]
    return x


def f64(x):
    return float(x)


def make_case(name, kind, d, N, variance, ls, noise, mean_const, seed, M=6, q=3, S=4, F=5, B=2):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, d))
    Y = rng.normal(size=N)
    # candidates: random, one exactly AT a training input, one a hair away, one far outside
    Xq = rng.uniform(size=(M, d))
    Xq[0] = X[0]
    if N > 1:
        Xq[1] = X[1] + 1e-7
    Xq[-1] = 5.0 + rng.uniform(size=d)  # far field: EI underflows, var -> variance
    ls = [float(v) for v in np.broadcast_to(np.asarray(ls, dtype=float), (d,))]

    K = mp.zeros(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = kern(kind, variance, ls, X[i], X[j])
        K[i, i] += mpf(noise)
    L = chol(K)
    err = [mpf(Y[i]) - mpf(mean_const) for i in range(N)]
    alpha = bsub(L, fsub(L, err))

    def post(xs):
        """mean list, full covariance matrix at the list of points xs."""
        n = len(xs)
        A = []
        means = []
        for x in xs:
            ks = [kern(kind, variance, ls, X[i], x) for i in range(N)]
            A.append(fsub(L, ks))
            means.append(sum(ks[i] * alpha[i] for i in range(N)) + mpf(mean_const))
        cov = mp.zeros(n, n)
        for a in range(n):
            for b_ in range(n):
                cov[a, b_] = kern(kind, variance, ls, xs[a], xs[b_]) - sum(
                    A[a][i] * A[b_][i] for i in range(N))
        return means, cov

    # predict at candidates
    mean, var_raw = [], []
    for x in Xq:
        m, c = post([x])
        mean.append(m[0])
        var_raw.append(c[0, 0])
    var = [max(v, mp.mpf("1e-12")) for v in var_raw]
    # eta = min posterior mean at the training inputs
    tm = [post([x])[0][0] for x in X]
    eta = min(tm)
    ei, pi_, lcb, aei = [], [], [], []
    for m, v in zip(mean, var):
        sd = mp.sqrt(v)
        z = (eta - m) / sd
        ei.append((eta - m) * mp.ncdf(z) + sd * mp.npdf(z))
        pi_.append(mp.ncdf(z))
        lcb.append(-(m - mp.mpf("1.96") * sd))
        aei.append(ei[-1] * (1 - mp.sqrt(mpf(noise)) / mp.sqrt(mpf(noise) + v)))

    # cross-covariance between the first 2 and the next 3 candidates (models.py:188-254): a block of
    # the raw (unclipped) joint posterior covariance of their union
    n1, n2 = 2, min(3, len(Xq) - 2)
    _, cu = post([Xq[a] for a in range(n1 + n2)])
    cov12 = [[f64(cu[a, n1 + b_]) for b_ in range(n2)] for a in range(n1)]

    # joint posterior for G=2 groups of q points, + qEI given eps
    G = 2
    Xg = rng.uniform(size=(G, q, d))
    if N > 1:
        Xg[0, 0] = 0.5 * (X[0] + X[1])
    eps = rng.normal(size=(q, S))
    jm, jc, qei = [], [], []
    jitter = mp.mpf("1e-6")
    for g in range(G):
        m, c = post([Xg[g, a] for a in range(q)])
        for a in range(q):
            c[a, a] = max(c[a, a], mp.mpf("1e-12"))
        jm.append([f64(v) for v in m])
        jc.append([[f64(c[a, b_]) for b_ in range(q)] for a in range(q)])
        cj = c.copy()
        for a in range(q):
            cj[a, a] += jitter
        Lq = chol(cj)
        acc = mp.mpf(0)
        for s in range(S):
            smp = [m[a] + sum(Lq[a, k] * mpf(eps[k, s]) for k in range(a + 1)) for a in range(q)]
            acc += max(eta - min(smp), mp.mpf(0))
        qei.append(f64(acc / S))

    # decoupled trajectory given (W, b, w, xi)
    Wf = rng.normal(size=(F, d))
    bf = rng.uniform(0, 2 * np.pi, size=F)
    w = rng.normal(size=(F, B))
    xi = rng.normal(size=(N, B))

    def phi(x):
        c = mp.sqrt(2 * mpf(variance) / F)
        return [c * mp.cos(sum(mpf(x[k]) / mpf(ls[k]) * mpf(Wf[f, k]) for k in range(d)) + mpf(bf[f]))
                for f in range(F)]

    phiZ = [phi(X[i]) for i in range(N)]
    vmat = []
    for bb in range(B):
        diff = [err[i] + mp.sqrt(mpf(noise)) * mpf(xi[i, bb])
                - sum(phiZ[i][f] * mpf(w[f, bb]) for f in range(F)) for i in range(N)]
        vmat.append(bsub(L, fsub(L, diff)))
    traj = []
    for x in Xq:
        ph = phi(x)
        ks = [kern(kind, variance, ls, X[i], x) for i in range(N)]
        traj.append([f64(sum(ph[f] * mpf(w[f, bb]) for f in range(F))
                         + sum(ks[i] * vmat[bb][i] for i in range(N)) + mpf(mean_const))
                     for bb in range(B)])

    # greedy batches (acquisition/function/greedy_batch.py): local penalizers around the q points of group 1
    # with given radius / scale (soft :341-354, hard :376-389), and the posterior conditioned additionally on
    # fantasised observations there (what _fantasized_model predicts, :630-773) = a refit on N + q points
    pend = [Xg[1, a] for a in range(q)]
    pen_r = rng.uniform(0.05, 0.4, size=q)
    pen_s = rng.uniform(0.05, 0.3, size=q)
    yf = rng.normal(size=q)
    pen_soft, pen_hard = [], []
    for x in Xq:
        ps, ph_ = mp.mpf(1), mp.mpf(1)
        for a in range(q):
            dist = mp.sqrt(sum((mpf(x[k]) - mpf(pend[a][k])) ** 2 for k in range(d)))
            ps *= mp.ncdf((dist - mpf(pen_r[a])) / mpf(pen_s[a]))
            u = dist / (mpf(pen_r[a]) + mpf(pen_s[a]))
            ph_ *= (u ** -5 + 1) ** (mp.mpf(-1) / 5) if dist > 0 else mp.mpf(0)
        pen_soft.append(f64(ps))
        pen_hard.append(f64(ph_))
    Xf = [X[i] for i in range(N)] + pend
    Nf = N + q
    Kf = mp.zeros(Nf, Nf)
    for i in range(Nf):
        for j in range(Nf):
            Kf[i, j] = kern(kind, variance, ls, Xf[i], Xf[j])
        Kf[i, i] += mpf(noise)
    Lf = chol(Kf)
    errf = err + [mpf(yf[a]) - mpf(mean_const) for a in range(q)]
    alphaf = bsub(Lf, fsub(Lf, errf))
    fant_mean, fant_var = [], []
    for x in Xq:
        ks = [kern(kind, variance, ls, Xf[i], x) for i in range(Nf)]
        A_ = fsub(Lf, ks)
        fant_mean.append(f64(sum(ks[i] * alphaf[i] for i in range(Nf)) + mpf(mean_const)))
        fant_var.append(f64(kern(kind, variance, ls, x, x) - sum(a_ * a_ for a_ in A_)))

    # entropy search (acquisition/function/entropy.py): MES (:195-214) and GIBBON's quality (:479-500) and repulsion
    # (:580-619) terms at the candidates, from the exact posterior, with given min-value samples
    S_ent = 4
    ent_samples = [float(f64(eta)) - abs(float(t)) * 0.3 for t in rng.normal(size=S_ent)]
    mes, gq, grep = [], [], []
    for idx, (m, v) in enumerate(zip(mean, var)):
        sd = max(mp.sqrt(v), mp.mpf("1e-8"))
        rho2 = v / (v + mpf(noise))
        acc_m, acc_g = mp.mpf(0), mp.mpf(0)
        for smp in ent_samples:
            u = (mpf(smp) - m) / sd
            lmc = mp.log(mp.ncdf(-u))
            r = mp.npdf(u) / mp.ncdf(-u)
            acc_m += -u * r / 2 - lmc
            acc_g += mp.log(1 + rho2 * r * (u - r))
        mes.append(f64(acc_m / S_ent))
        gq.append(f64(-acc_g / (2 * S_ent)))
        # repulsion by the reference's block-determinant formula, from the joint posterior of [x; pending]
        _, cj = post([Xq[idx]] + pend)
        Bm = mp.zeros(q, q)
        for a in range(q):
            for b_ in range(q):
                Bm[a, b_] = cj[1 + a, 1 + b_]
            Bm[a, a] += mpf(noise)
        Lb = chol(Bm)
        Avec = [cj[0, 1 + a] for a in range(q)]
        LiA = fsub(Lb, Avec)
        yvar = v + mpf(noise)
        vdet = yvar - sum(t * t for t in LiA)
        grep.append(f64((mp.log(vdet) - mp.log(yvar)) / 2 / (q * q)))
        # ... which is the conditioned (fantasised) variance + noise: the identity the engine's twin relies on
        if var_raw[idx] > mp.mpf("1e-12"):
            assert abs(vdet - (mpf(fant_var[idx]) + mpf(noise))) < mp.mpf("1e-13") * (1 + abs(vdet)), (name, idx)

    return {
        "ent_samples": ent_samples, "mes": mes, "gibbon_quality": gq, "gibbon_repulsion": grep,
        "pen_radius": pen_r.tolist(), "pen_scale": pen_s.tolist(), "pen_soft": pen_soft, "pen_hard": pen_hard,
        "fant_y": yf.tolist(), "fant_mean": fant_mean, "fant_var_raw": fant_var,
        "name": name, "kind": kind, "d": d, "N": N, "variance": variance, "lengthscales": ls,
        "noise": noise, "mean_const": mean_const,
        "X": X.tolist(), "Y": Y.tolist(), "Xq": Xq.tolist(),
        "L": [[f64(L[i, j]) for j in range(N)] for i in range(N)],
        "alpha": [f64(a) for a in alpha],
        "mean": [f64(m) for m in mean], "var_raw": [f64(v) for v in var_raw],
        "var": [f64(v) for v in var], "eta": f64(eta),
        "ei": [f64(v) for v in ei], "pi": [f64(v) for v in pi_], "nlcb": [f64(v) for v in lcb],
        "aei": [f64(v) for v in aei], "cov12": cov12,
        "Xg": Xg.tolist(), "eps": eps.tolist(), "joint_mean": jm, "joint_cov": jc, "qei": qei,
        "jitter": 1e-6,
        "rff_W": Wf.tolist(), "rff_b": bf.tolist(), "traj_w": w.tolist(), "traj_xi": xi.tolist(),
        "traj_v": [[f64(vmat[bb][i]) for bb in range(B)] for i in range(N)],
        "traj": traj,
    }


def make_wide_qei_case(name, kind, d, N, variance, ls, noise, mean_const, seed, q, S=8, G=2, keep_cov=True):
    """Batch Monte-Carlo EI at the group sizes the engine's one-wave tail is instantiated for (QP = 16 / 32 / 64 and
    BASELINE config 4's q = 50): joint mean / covariance, the reparametrised samples mean + chol(cov + 1e-6 I) eps and
    qEI = mean_S max(eta - min_q sample, 0) in 50-digit arithmetic (function.py:1181-1186, sampler.py:276-287).  eta = the
    median of the groups' posterior means, so that every value is O(0.1 .. 1) -- at the reference's eta = min training
    mean random groups have qEI = 0 exactly and a comparison says nothing."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, d))
    Y = rng.normal(size=N)
    ls = [float(v) for v in np.broadcast_to(np.asarray(ls, dtype=float), (d,))]
    K = mp.zeros(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = kern(kind, variance, ls, X[i], X[j])
        K[i, i] += mpf(noise)
    L = chol(K)
    err = [mpf(Y[i]) - mpf(mean_const) for i in range(N)]
    alpha = bsub(L, fsub(L, err))
    Xg = rng.uniform(size=(G, q, d))
    Xg[0, 0] = X[0]          # a point AT a training input inside a group
    eps = rng.normal(size=(q, S))
    jitter = mp.mpf("1e-6")
    means, covs = [], []
    for g in range(G):
        A, m = [], []
        for a in range(q):
            ks = [kern(kind, variance, ls, X[i], Xg[g, a]) for i in range(N)]
            A.append(fsub(L, ks))
            m.append(sum(ks[i] * alpha[i] for i in range(N)) + mpf(mean_const))
        c = mp.zeros(q, q)
        for a in range(q):
            for b_ in range(a + 1):
                c[a, b_] = c[b_, a] = kern(kind, variance, ls, Xg[g, a], Xg[g, b_]) - sum(A[a][i] * A[b_][i] for i in range(N))
            c[a, a] = max(c[a, a], mp.mpf("1e-12"))
        means.append(m)
        covs.append(c)
    flat = sorted(v for m in means for v in m)
    eta = (flat[len(flat) // 2 - 1] + flat[len(flat) // 2]) / 2 if len(flat) % 2 == 0 else flat[len(flat) // 2]
    samples, qei = [], []
    for g in range(G):
        cj = covs[g].copy()
        for a in range(q):
            cj[a, a] += jitter
        Lq = chol(cj)
        acc, smp_g = mp.mpf(0), []
        for sidx in range(S):
            smp = [means[g][a] + sum(Lq[a, k] * mpf(eps[k, sidx]) for k in range(a + 1)) for a in range(q)]
            smp_g.append([f64(v) for v in smp])
            acc += max(eta - min(smp), mp.mpf(0))
        samples.append(smp_g)
        qei.append(f64(acc / S))
    out = {"name": name, "kind": kind, "d": d, "N": N, "q": q, "S": S, "variance": variance, "lengthscales": ls,
           "noise": noise, "mean_const": mean_const, "X": X.tolist(), "Y": Y.tolist(), "Xg": Xg.tolist(),
           "eps": eps.tolist(), "eta": f64(eta), "jitter": 1e-6,
           "joint_mean": [[f64(v) for v in m] for m in means], "samples": samples, "qei": qei}
    if keep_cov:  # (q^2 numbers per group: kept where it is small, and for one q = 50 case)
        out["joint_cov"] = [[[f64(c[a, b_]) for b_ in range(q)] for a in range(q)] for c in covs]
    return out


def main_wide(out_path):
    cases, sid = [], 900
    for kind, d, N in (("matern52", 2, 8), ("rbf", 6, 8)):
        for q in (9, 17, 33, 50):
            sid += 1
            ls = [0.2 * np.sqrt(d) * (1.0 + 0.3 * k) for k in range(d)]
            cases.append(make_wide_qei_case(f"{kind}_d{d}_q{q}", kind, d, N, 1.3, ls, 1e-2, 0.1, sid, q,
                                            keep_cov=q <= 17 or (q == 50 and kind == "matern52")))
    with open(out_path, "w") as f:
        json.dump({"generator": "oracle/make_goldens.py --wide-qei", "mp_dps": mp.mp.dps, "cases": cases}, f)
    print(f"wrote {len(cases)} wide-qEI cases to {out_path} ({os.path.getsize(out_path)} bytes)")


def main(out_path):
    cases = []
    sid = 100
    for kind in ("rbf", "matern52"):
        for (N, d) in ((1, 1), (2, 2), (5, 2), (8, 6), (8, 1), (5, 6)):
            for noise in (1e-7, 1e-3, 1e-1):
                sid += 1
                ls = [0.2 * np.sqrt(d) * (1.0 + 0.3 * k) for k in range(d)]  # ARD
                variance = 1.0 if noise != 1e-3 else 2.5
                cases.append(make_case(f"{kind}_N{N}_d{d}_s{noise:g}", kind, d, N, variance, ls,
                                       noise, 0.3 if noise == 1e-1 else 0.0, sid))
    for kind in ("matern12", "matern32"):
        sid += 1
        cases.append(make_case(f"{kind}_N5_d2", kind, 2, 5, 1.3, [0.3, 0.5], 1e-4, -0.2, sid))
    # tie case: two identical candidates (first index must win) is exercised in tests directly.
    with open(out_path, "w") as f:
        json.dump({"generator": "oracle/make_goldens.py", "mp_dps": mp.mp.dps, "cases": cases}, f)
    print(f"wrote {len(cases)} cases to {out_path} ({os.path.getsize(out_path)} bytes)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--wide-qei":
        main_wide(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                               "qei_wide_goldens.json"))
        sys.exit(0)
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
        "gp_goldens.json"))
