"""Development aid: cProfile of one greedy-batch acquire (LocalPenalization / Fantasizer, q = 10) at N.
usage: python tools/prof_greedy.py [N=4000] [lp|lphard|kb]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import objectives as OBJ
import trieste_amd.models as M
import trieste_amd.acquisition as A
from trieste_amd.data import Dataset
from trieste_amd.space import Box

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 8
what = sys.argv[2] if len(sys.argv) > 2 else "lp"
X, Y = OBJ.synthetic_problem(OBJ.ackley, d, N)
Y = Y[:, None]
space = Box([0.0] * d, [1.0] * d)
data = Dataset(X, Y)
model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
builder = {"lp": lambda: A.LocalPenalization(space), "lphard": lambda: A.LocalPenalization(space, penalizer=A.hard_local_penalizer),
           "kb": lambda: A.Fantasizer()}[what]()
rule = A.EfficientGlobalOptimization(builder, num_query_points=10)
rule.acquire_single(space, model, data)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); rule.acquire_single(space, model, data); ts.append((time.perf_counter() - t0) * 1e3)
print(what, "acquire ms:", " ".join(f"{t:.0f}" for t in ts))
pr = cProfile.Profile(); pr.enable(); rule.acquire_single(space, model, data); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
