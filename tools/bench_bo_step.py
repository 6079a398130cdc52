"""Host-level latency of one BO step's model side at N (development aid): update (append), optimize (prior draws +
L-BFGS-B).  usage: python tools/bench_bo_step.py [N]      (TGP_NO_DAG=1 forces the recursion of round 2)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.optimize  # noqa: F401  (a ~190 ms import the first optimize() of a process would otherwise pay)
from trieste_amd import objectives as O
import trieste_amd.models as M
from trieste_amd.data import Dataset
from trieste_amd.space import Box

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
X, Y = O.synthetic_problem(O.ackley, d, N + 1)
Y = Y[:, None]
space = Box([0.0] * d, [1.0] * d)
data = Dataset(X[:N], Y[:N])
model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
model.update(data)
newd = Dataset(X, Y)
for rep in range(2):
    model.update(data)
    t0 = time.perf_counter(); model.update(newd); t1 = time.perf_counter()
    res = model.optimize(newd); t2 = time.perf_counter()
    print(f"N={N}: update(append 1 row) {1e3*(t1-t0):.2f} ms, optimize {1e3*(t2-t1):.0f} ms (nfev={res.nfev}), "
          f"workers={model.MAX_PARALLEL_EVALUATIONS}" + ("  [first call of the process: allocations, task plans, scratch]" if rep == 0 else ""),
          flush=True)
for rep in range(2):
    t0 = time.perf_counter(); model.find_best_model_initialization(90); t1 = time.perf_counter()
    print(f"  find_best_model_initialization(90) batched (tgp_nlml_trial_batch) after the append: N + 1 rows, padded to the next 256: {1e3*(t1-t0):.0f} ms", flush=True)
type(model).BATCHED_TRIALS = False
for w in (1, 3):
    model.PERSISTENT_UPDATE_WORKERS = w
    t0 = time.perf_counter(); model.find_best_model_initialization(90); t1 = time.perf_counter()
    print(f"  find_best_model_initialization(90) one by one, workers={w}: {1e3*(t1-t0):.0f} ms", flush=True)
type(model).BATCHED_TRIALS = True
# a COLD fit: a fresh model from build_gpr defaults (first optimize: allocations, plans, worker engines)
cold = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
t0 = time.perf_counter(); res = cold.optimize(data); t1 = time.perf_counter()
print(f"  COLD optimize (fresh model, build_gpr defaults): {1e3*(t1-t0):.0f} ms (nfev={res.nfev})", flush=True)
t0 = time.perf_counter(); cold.find_best_model_initialization(90); t1 = time.perf_counter()
print(f"  find_best_model_initialization(90) batched at N = {N} exactly (Npad = {N}): {1e3*(t1-t0):.0f} ms", flush=True)
cold2 = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
t0 = time.perf_counter(); res = cold2.optimize(data); t1 = time.perf_counter()
print(f"  second fresh model, same process (allocator warm): {1e3*(t1-t0):.0f} ms (nfev={res.nfev})", flush=True)
