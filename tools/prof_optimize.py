"""Development aid: cProfile of GaussianProcessRegression.optimize (warm) at N.   usage: python tools/prof_optimize.py [N=1024]"""
import sys, os, time, cProfile, pstats
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
ts = []
for rep in range(4):
    model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
    t0 = time.perf_counter(); res = model.optimize(data); ts.append(((time.perf_counter() - t0) * 1e3, res.nfev))
print("cold optimize ms (nfev):", ts)
model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
pr = cProfile.Profile(); pr.enable(); model.optimize(data); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
