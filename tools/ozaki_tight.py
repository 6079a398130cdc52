"""Error study for the i8x4 'tight scales + a-posteriori repair' design.

Emulates the digit-plane product with row scales S_i = f max_k |W_ik| and S' = f s_f^2 (f = 2: round 2/3 kernels,
f = 1 + 2^-7: tight), measures |d var| against 80-bit sums, and the per-candidate rms model
    rms_j = 2 * 2^-32.8 S' sqrt(sum_i c_ij^2 S_i^2 (i + 1))
usage: python tools/ozaki_tight.py [N=4096] [noise=1e-2] [M=2048]
"""
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ozaki_bound import digits, matern52  # noqa: E402
from trieste_amd import objectives as OBJ  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    noise = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    NS, d = 4, 8
    X, _ = OBJ.synthetic_problem(OBJ.ackley, d, N)
    ls = OBJ.default_lengthscales(d)
    K = matern52(X, X, ls)
    K[np.diag_indices(N)] += noise
    W = sl.solve_triangular(sl.cholesky(K, lower=True), np.eye(N), lower=True)
    rng = np.random.default_rng(5678)
    Xq = rng.uniform(size=(M, d))
    Xq[:32] = X[:32]
    Xq[32:64] = X[32:64] + 1e-5 * rng.standard_normal((32, d))
    Xq[64:96] = X[64:96] + 1e-2 * rng.standard_normal((32, d))
    Ks = matern52(X, Xq, ls)
    c_ref = W.astype(np.longdouble) @ Ks.astype(np.longdouble)
    var_ref = np.maximum((1.0 - np.sum(c_ref * c_ref, axis=0)).astype(np.float64), 1e-12)
    floor = min(64 * np.finfo(float).eps * (1 + N / noise), 1e-6)
    tol = 1e-5 * np.abs(var_ref) + floor
    print(f"N={N} noise={noise:g} M={M} floor={floor:.3g} var quantiles {np.quantile(var_ref, [0, .01, .5, 1])}")
    for f in (2.0, 1.0 + 2.0 ** -7):
        Sa = f * np.max(np.abs(W), axis=1, keepdims=True)
        Sb = f * 1.0
        da = digits(np.rint(W / Sa * 2.0 ** 31).astype(np.int64), NS)
        db = digits(np.rint(Ks / Sb * 2.0 ** 31).astype(np.int64), NS)
        assert all(x.min() >= -128 and x.max() <= 127 for x in da + db), [(x.min(), x.max()) for x in da + db]
        acc = [np.zeros((N, M)) for _ in range(NS)]
        for s in range(NS):
            for t in range(NS):
                if s + t <= NS - 1:
                    acc[s + t] += da[s].astype(np.float64) @ db[t].astype(np.float64)
        c = Sa * Sb * sum(acc[g] * 2.0 ** (-14 - 8 * g) for g in range(NS))
        var = np.maximum(1.0 - np.sum(c * c, axis=0), 1e-12)
        err = np.abs(var - var_ref)
        wi = (Sa[:, 0] ** 2) * (np.arange(N) + 1.0)
        rms = 2.0 * 2.0 ** -32.8 * Sb * np.sqrt((c * c * wi[:, None]).sum(0))
        apriori = 2.0 * 2.0 ** -32.8 * Sb * Sa.max() * np.sqrt(N)
        print(f" scale factor {f:.5f}: max err/tol {np.max(err / tol):.3g}  median {np.median(err / tol):.3g};  "
              f"err/rms: max {np.max(err / rms):.3g} rms {np.sqrt(np.mean((err / rms) ** 2)):.3g};  "
              f"rms/tol: max {np.max(rms / tol):.3g} median {np.median(rms / tol):.3g}; a-priori/rms median {np.median(apriori / rms):.3g}")
        for k in (3, 4, 5, 6):
            print(f"   flagged at k={k} (k rms > tol): {np.mean(k * rms > tol):.4f}")


if __name__ == "__main__":
    main()
