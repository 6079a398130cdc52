"""Development aid: qEI value-and-gradient (tgp_joint_forward + host adjoint + tgp_joint_vjp) and an EGO acquire of a joint
batch with the default optimizer (L-BFGS-B over the q x d batch since round 6; random search before).
usage: python tools/bench_qei_grad.py [N=2048] [q=5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import objectives as OBJ
import trieste_amd.models as M
import trieste_amd.acquisition as A
from trieste_amd.data import Dataset
from trieste_amd.space import Box

N, q, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 5, 6
X, Y = OBJ.synthetic_problem(OBJ.hartmann_6, d, N)
space = Box([0.0] * d, [1.0] * d)
data = Dataset(X, Y[:, None])
model = M.GaussianProcessRegression(M.build_gpr(data, space, likelihood_variance=1e-2))
eng = model.engine
rng = np.random.default_rng(0)


def t(f, n=5):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


fn = A.BatchMonteCarloExpectedImprovement(512).prepare_acquisition_function(model, dataset=data)
for G in (10, 60, 300):
    G = min(G, 2048 // q)
    xs = rng.uniform(size=(G, q, d))
    gm, gc = rng.normal(size=(G, q)), rng.normal(size=(G, q, q))
    eng.set_variant(1024)   # bit 10: small calls through the joint kernel (rounds 1 - 5)
    kern_pj, kern_qei = t(lambda: eng.predict_joint(xs)), t(lambda: fn(xs))
    eng.set_variant(0)
    print(f"N={N} q={q} G={G}: joint_forward {t(lambda: eng.joint_forward(xs)):.2f} ms, predict_joint {t(lambda: eng.predict_joint(xs)):.2f} ms "
          f"(through the joint kernel: {kern_pj:.2f} ms; tgp_qei {kern_qei:.2f} ms), joint_vjp {t(lambda: eng.joint_vjp(xs, gm, gc)):.2f} ms, "
          f"qEI value_and_gradient {t(lambda: fn.value_and_gradient(xs)):.2f} ms, qEI value (tgp_qei) {t(lambda: fn(xs)):.2f} ms")
for name, opt in (("default (L-BFGS-B on the flattened batch)", None),
                  ("random search, 10^5 batches", A.generate_random_search_optimizer(100000))):
    rule = A.EfficientGlobalOptimization(A.BatchMonteCarloExpectedImprovement(512), num_query_points=q, **({"optimizer": opt} if opt else {}))
    rule.acquire_single(space, model, data)
    t0 = time.perf_counter(); pts = rule.acquire_single(space, model, data); ms = (time.perf_counter() - t0) * 1e3
    val = float(np.asarray(rule.acquisition_function(pts[None]))[0, 0])
    print(f"EGO qEI q={q}, {name}: acquire {ms:.0f} ms, qEI of the returned batch {val:.6g}")
