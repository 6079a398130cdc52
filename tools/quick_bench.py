"""Quick timing of update + sweep at a few sizes (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trieste_amd import objectives as O  # seeded synthetic problems (product side)
from trieste_amd.engine import GPEngine

def run(kind, d, N, M, noise=1e-2, variant=0, reps=3):
    obj = O.hartmann_6 if d == 6 else O.ackley
    X, Y = O.synthetic_problem(obj, d, N)
    eng = GPEngine(d, kind)
    eng.set_variant(variant)
    eng.set_hyper(1.0, O.default_lengthscales(d), noise, float(Y.mean()))
    t0 = time.perf_counter(); eng.set_data(X, Y); t1 = time.perf_counter()
    eng.set_data(X, Y); t2 = time.perf_counter()
    eta = eng.eta()
    Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
    eng.use_torch_stream()
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t3 = time.perf_counter()
        val, idx, x = eng.acq_argmax("ei", eta, Xq)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        ms, _ = eng.last_kernel_ms()
        best = ms if best is None else min(best, ms)
    Npad = (N + 127) // 128 * 128
    flops = M * (float(N) * N)            # algorithmic: N^2 per candidate
    print(f"{kind} d={d} N={N} M={M} var={variant}: update {1e3*(t2-t1):.1f} ms (first {1e3*(t1-t0):.1f}); "
          f"sweep kernel {best:.2f} ms wall {1e3*(t4-t3):.2f} ms -> {M/best*1e3:.3e} cand/s, "
          f"{flops/best*1e-9:.2f} TFLOP/s algorithmic; best EI {val:.4e} @ {idx}", flush=True)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for v in (1, 0):
        run("matern52", 8, 4096, 1 << 17, variant=v)
        run("matern52", 8, 4096, 1 << 20, reps=2, variant=v)
        run("rbf", 6, 1024, 1 << 20, reps=2, variant=v)
    run("matern52", 6, 2048, 1 << 18)
    run("matern52", 16, 8192, 1 << 17, reps=2)
