"""Where a COLD fit spends its time (development aid): fresh models from build_gpr defaults at N; per model the prior
draws (find_best_model_initialization(10)) and the L-BFGS-B phase timed separately, plus the per-call cost of the loss."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import trieste_amd.models as M
from trieste_amd import objectives as O
from trieste_amd.data import Dataset
from trieste_amd.space import Box

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
X, Y = O.synthetic_problem(O.ackley, d, N)
data = Dataset(X, Y[:, None])
space = Box([0.0] * d, [1.0] * d)
for rep in range(4):
    t0 = time.perf_counter()
    model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
    t1 = time.perf_counter()
    model.find_best_model_initialization(10, seed=rep)
    t2 = time.perf_counter()
    calls = []
    orig = model._loss_at

    def timed(*a, **k):
        s = time.perf_counter()
        out = orig(*a, **k)
        calls.append(time.perf_counter() - s)
        return out

    model._loss_at = timed
    model._num_kernel_samples = 0
    res = model.optimize(data)
    t3 = time.perf_counter()
    print(f"fresh model {rep}: construct {1e3*(t1-t0):.1f} ms, 10 prior draws {1e3*(t2-t1):.1f} ms, L-BFGS-B {1e3*(t3-t2):.1f} ms "
          f"(nfev={res.nfev}; loss calls {len(calls)}: first {1e3*calls[0]:.1f} ms, median {1e3*np.median(calls):.2f} ms, max {1e3*max(calls):.1f} ms)",
          flush=True)
    import gc
    t4 = time.perf_counter()
    del model, timed, orig
    gc.collect()
    print(f"   releasing the model (engine buffers + {10} x 3 N^2 doubles of batch scratch): {1e3*(time.perf_counter()-t4):.1f} ms", flush=True)
