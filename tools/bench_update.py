"""Timing of `update` (K assembly + Cholesky + inverse + alpha) and of NLML + gradient (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trieste_amd import objectives as O  # seeded synthetic problems (product side)
from trieste_amd.engine import GPEngine

def run(kind, d, N, reps=5):
    X, Y = O.synthetic_problem(O.ackley, d, N)
    eng = GPEngine(d, kind)
    eng.set_variant(int(os.environ.get('TGP_VARIANT', '0')))   # (e.g. 256: the persistent kernel's plan without round 6's split)
    eng.set_hyper(1.0, O.default_lengthscales(d), 1e-2, float(Y.mean()))
    eng.set_data(X, Y)
    ts, tn = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.set_data(X, Y); t1 = time.perf_counter()
        eng.nlml(); t2 = time.perf_counter()
        ts.append(t1 - t0); tn.append(t2 - t1)
    L, W, alpha = eng.get_factor()
    eL = np.abs(W @ L - np.eye(N)).max()  # self-consistency only; parity lives in tests/
    ea = np.abs(L @ (L.T @ alpha) - (Y - float(Y.mean()))).max()
    t3 = time.perf_counter(); eta = eng.eta(); t4 = time.perf_counter()
    print(f"{kind} d={d} N={N}: update {1e3*min(ts):.2f} ms, nlml+grad {1e3*min(tn):.2f} ms, eta {1e3*(t4-t3):.2f} ms; "
          f"|W L - I| {eL:.1e} |K alpha - err| {ea:.1e}", flush=True)

if __name__ == "__main__":
    for N in [int(a) for a in sys.argv[1:]] or (256, 1024, 2048, 4096, 8192):
        run("matern52", 8, N)
