"""Development aid: checksums of the fused sweep's outputs (mean, variance, EI, arg-max) over kernel kinds and dimensions
at a size that takes the LDS-DMA kernel (>= 4 x #CU candidate blocks).  Run under two builds (TGP_LIB=...) and diff the
output: a change that only moves WHEN instructions issue must leave every line identical."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trieste_amd import objectives as O
from trieste_amd.engine import GPEngine

def h(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]

CONFIGS = [(k, d, N) for k in ("rbf", "matern12", "matern32", "matern52") for d, N in ((2, 600), (6, 1000), (8, 2100), (16, 600))]
CONFIGS += [("matern52", 8, 4096), ("rbf", 4, 1024), ("matern32", 3, 300)]   # the headline size; dp = 4
for kind, d, N in CONFIGS:
    if True:
        f = O.hartmann_6 if d == 6 else O.ackley
        X, Y = O.synthetic_problem(f, d, N)
        eng = GPEngine(d, kind); eng.set_hyper(1.3, O.default_lengthscales(d), 1e-2, float(Y.mean())); eng.set_data(X, Y)
        Xq = eng.sample_box(99, 0, 140000, 0.0, 1.0)
        mean, var = eng.predict(Xq)
        ei = eng.acq_values("ei", eng.eta(), Xq)
        v, i, _ = eng.acq_argmax("ei", eng.eta(), Xq)
        print(kind, d, N, h(mean), h(var), h(ei), repr(float(v)), int(i), flush=True)
