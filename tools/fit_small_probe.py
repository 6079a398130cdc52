"""Development aid (round 6): find_best_model_initialization(90) and optimize() at a small N -- run under rocprofv3 --kernel-trace --stats to
see where a batched launch group's time goes.   usage: python tools/fit_small_probe.py [N=1024]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.optimize  # noqa: F401
from trieste_amd import objectives as O
import trieste_amd.models as M
from trieste_amd.data import Dataset
from trieste_amd.space import Box

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 8
X, Y = O.synthetic_problem(O.ackley, d, N)
Y = Y[:, None]
space = Box([0.0] * d, [1.0] * d)
data = Dataset(X, Y)
model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
model.update(data)
model.find_best_model_initialization(90)
for rep in range(3):
    t0 = time.perf_counter(); model.find_best_model_initialization(90); t1 = time.perf_counter()
    print(f"N={N}: find_best_model_initialization(90) {1e3 * (t1 - t0):.2f} ms", flush=True)
for rep in range(2):
    t0 = time.perf_counter(); res = model.optimize(data); t1 = time.perf_counter()
    print(f"N={N}: optimize {1e3 * (t1 - t0):.1f} ms (nfev={res.nfev})", flush=True)
