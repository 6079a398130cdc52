"""End-to-end latency of the host-level operations of one BO step at N=4096, d=8 (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from trieste_amd.engine import GPEngine
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
X, Y = O.synthetic_problem(O.ackley, d, N)
eng = GPEngine(d, "matern52"); eng.set_hyper(1.0, O.default_lengthscales(d), 1e-2, float(Y.mean())); eng.set_data(X, Y)
eta = eng.eta()
rng = np.random.default_rng(0)
for P in (80, 128, 500):
    Xp = rng.uniform(size=(P, d))
    eng.acq_value_grad("ei", eta, Xp)
    t0 = time.perf_counter()
    for _ in range(20): eng.acq_value_grad("ei", eta, Xp)
    print(f"acq_value_grad P={P}: {(time.perf_counter()-t0)/20*1e3:.3f} ms", flush=True)
for M in (8000, 16000, 100000):
    Xq = rng.uniform(size=(M, d))
    eng.acq_topk("ei", eta, Xq, 80)
    t0 = time.perf_counter()
    for _ in range(5): eng.acq_topk("ei", eta, Xq, 80)
    print(f"acq_topk M={M} k=80: {(time.perf_counter()-t0)/5*1e3:.3f} ms", flush=True)
    t0 = time.perf_counter()
    for _ in range(5): eng.acq_argmax("ei", eta, Xq)
    print(f"acq_argmax M={M}: {(time.perf_counter()-t0)/5*1e3:.3f} ms", flush=True)
X1, X2 = rng.uniform(size=(50, d)), rng.uniform(size=(2000, d))
eng.cov_between(X1, X2); t0 = time.perf_counter(); eng.cov_between(X1, X2); print(f"cov_between 50 x 2000: {(time.perf_counter()-t0)*1e3:.3f} ms")
