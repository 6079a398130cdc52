"""End-to-end latency of the host-level operations of one BO step at N=4096, d=8 (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import objectives as O  # seeded synthetic problems (product side)
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

# one full BO step through the host API: model.update (append) + model.optimize (MAP fit) + rule.acquire
import trieste_amd.models as M
from trieste_amd.acquisition import EfficientGlobalOptimization
from trieste_amd.data import Dataset
from trieste_amd.space import Box
for n in (1000, N):
    Xn, Yn = X[:n], Y[:n, None] if Y.ndim == 1 else Y[:n]
    space = Box([0.0] * d, [1.0] * d)
    data = Dataset(Xn, Yn)
    model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
    rule = EfficientGlobalOptimization()
    rule.acquire_single(space, model, dataset=data)
    newx = rng.uniform(size=(1, d)); newd = data + Dataset(newx, np.array([[0.1]]))
    t0 = time.perf_counter(); model.update(newd); t1 = time.perf_counter()
    res = model.optimize(newd); t2 = time.perf_counter()
    rule.acquire_single(space, model, dataset=newd); t3 = time.perf_counter()
    nfev = getattr(res, "nfev", None)
    print(f"BO step at N={n}: update {1e3*(t1-t0):.1f} ms, optimize {1e3*(t2-t1):.0f} ms (nfev={nfev}), acquire {1e3*(t3-t2):.1f} ms", flush=True)
