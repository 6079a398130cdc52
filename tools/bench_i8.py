"""Kernel time of the headline sweep (N = 4096, d = 8, Matern-5/2, 2^20 candidates) per arithmetic, for A/B builds
(development aid).  usage: TGP_LIB=tools/exp/libtgp_x.so python tools/bench_i8.py [f64 i8x4 i8x5 auto ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trieste_amd import objectives as O
from trieste_amd.engine import GPEngine

N, d, M = 4096, 8, 1 << 20
X, Y = O.synthetic_problem(O.ackley, d, N)
eng = GPEngine(d, "matern52")
eng.use_torch_stream()
eng.set_variant(int(os.environ.get("TGP_VARIANT", "0")))   # e.g. 64: AUTO repairs through the SPLIT sweep only
eng.set_hyper(1.0, O.default_lengthscales(d), 1e-2, float(Y.mean()))
eng.set_data(X, Y)
eta = eng.eta()
Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
for prec in (sys.argv[1:] or ["i8x4", "i8x5", "auto", "f64"]):
    eng.set_precision(prec)
    eng.acq_argmax("ei", eta, Xq)
    ms = []
    for _ in range(3):
        v, i, _ = eng.acq_argmax("ei", eta, Xq)
        ms.append(eng.last_kernel_ms()[0])
    print(f"{os.environ.get('TGP_LIB', 'libtgp.so'):32s} {prec:5s}: kernel {min(ms):8.2f} ms (median {np.median(ms):.2f})  "
          f"{M / min(ms) * 1e3:.3e} cand/s   best {v:.6e} @ {i}   {eng.get_precision()}", flush=True)
