import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trieste_amd import objectives as O  # seeded synthetic problems (product side)
from trieste_amd.engine import GPEngine
d, N, M = 8, 4096, 1 << 18
X, Y = O.synthetic_problem(O.ackley, d, N)
eng = GPEngine(d, "matern52")
eng.set_hyper(1.0, O.default_lengthscales(d), 1e-2, float(Y.mean()))
eng.set_data(X, Y)
eta = eng.eta()
Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
Xfar = eng.sample_box(5678, 0, M, 10.0, 11.0)
for name, v, xq in (("u16", 0, Xq), ("u16 skip-gen", (1 << 8), Xq), ("u16 no diagonal skipping", (8 << 8), Xq),
                    ("u16 far candidates (K* underflows to 0)", 0, Xfar), ("u16 no setprio", (16 << 8), Xq), ("u16", 0, Xq)):
    eng.set_variant(v)
    r0 = eng.acq_argmax("ei", eta, xq)
    r1 = eng.acq_argmax("ei", eta, xq)
    ms, _ = eng.last_kernel_ms()
    print(f"{name:42s}: {ms:8.2f} ms  -> {M*float(N)*N/ms*1e-9:6.2f} TF  best={r1[0]:.6e}@{r1[1]}", flush=True)
