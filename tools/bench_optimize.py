import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from trieste_amd import objectives as O  # seeded synthetic problems (product side)
import trieste_amd.models as M
from trieste_amd.data import Dataset
from trieste_amd.space import Box
N, d = 4096, 8
X, Y = O.synthetic_problem(O.ackley, d, N)
space = Box([0.0] * d, [1.0] * d)
data = Dataset(X, Y[:, None])
model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
for w in (8, 1, 4, 8, 16):
    M.GaussianProcessRegression.MAX_PARALLEL_EVALUATIONS = w
    t0 = time.perf_counter(); model.find_best_model_initialization(90, seed=1); t1 = time.perf_counter()
    print(f"workers={w}: find_best_model_initialization(90) {1e3*(t1-t0):.0f} ms", flush=True)
