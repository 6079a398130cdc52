"""Development aid: cProfile of one default EfficientGlobalOptimization().acquire_single at N (the bench's `acquire_ms`).
usage: python tools/prof_acquire.py [N=4096]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trieste_amd.models as M
from trieste_amd import objectives as O
from trieste_amd.acquisition import EfficientGlobalOptimization
from trieste_amd.data import Dataset
from trieste_amd.space import Box
import trieste_amd.acquisition.optimizer as _opt
_opt.LOCKSTEP_LBFGSB = os.environ.get('TGP_LOCKSTEP', '1') != '0'   # (0: scipy.optimize.minimize + a greenlet per start)

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
X, Y = O.synthetic_problem(O.ackley, d, N)
kern = M.Kernel(variance=1.0, lengthscales=O.default_lengthscales(d), kind="matern52")
model = M.GaussianProcessRegression(M.GPR(data=(X, Y[:, None]), kernel=kern, mean_function=M.Constant(float(Y.mean())), likelihood_variance=1e-2))
data = Dataset(X, Y[:, None])
rule = EfficientGlobalOptimization()
space = Box([0.0] * d, [1.0] * d)
rule.acquire_single(space, model, dataset=data)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); rule.acquire_single(space, model, dataset=data); ts.append((time.perf_counter() - t0) * 1e3)
print("acquire_single ms:", " ".join(f"{t:.1f}" for t in ts))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): rule.acquire_single(space, model, dataset=data)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
