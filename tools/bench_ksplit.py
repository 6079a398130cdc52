"""Development aid: tgp_acq_value_grad / tgp_predict at a handful of points under the k-split rule of the tall products
(TGP_KSPLIT_MAX, TGP_KSPLIT_TARGET: read once per process).  usage: python tools/bench_ksplit.py [N=4096]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import objectives as O
from trieste_amd.engine import GPEngine
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
X, Y = O.synthetic_problem(O.ackley, d, N)
eng = GPEngine(d, "matern52"); eng.set_hyper(1.0, O.default_lengthscales(d), 1e-2, float(Y.mean())); eng.set_data(X, Y)
eta = eng.eta()
rng = np.random.default_rng(0)
out = []
for P in (16, 80, 128, 500, 1024, 2048):
    Xp = rng.uniform(size=(P, d))
    for name, f in (("value_grad", lambda: eng.acq_value_grad("ei", eta, Xp)), ("predict", lambda: eng.predict(Xp))):
        f()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        out.append(f"{name} P={P}: {sorted(ts)[15]:.3f}")
print(f"N={N} max={os.environ.get('TGP_KSPLIT_MAX', 'library rule')} target={os.environ.get('TGP_KSPLIT_TARGET', 'library rule')} ms: " + "  ".join(out))
