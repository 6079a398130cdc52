"""rocprofv3 target: three `update`s + NLML/gradient evaluations at N=4096 (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trieste_amd import objectives as O  # seeded synthetic problems (product side)
from trieste_amd.engine import GPEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, Y = O.synthetic_problem(O.ackley, 8, N)
eng = GPEngine(8, "matern52"); eng.set_hyper(1.0, O.default_lengthscales(8), 1e-2, float(Y.mean()))
for _ in range(3):
    eng.set_data(X, Y); eng.nlml()
eng.eta()
