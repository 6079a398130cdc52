"""`update` latency at N with the persistent kernel on 1 / n of the compute units, alone and with n engines side by side
(development aid for tgp_set_update_concurrency)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import objectives as O
from trieste_amd.engine import GPEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, Y = O.synthetic_problem(O.ackley, 8, N)
def make(n):
    e = GPEngine(8, "matern52"); e.use_private_stream(); e.set_update_concurrency(n)
    e.set_hyper(1.0, O.default_lengthscales(8), 1e-2, float(Y.mean())); e.set_data(X, Y); return e
for n in (1, 2, 3):
    e = make(n)
    t0 = time.perf_counter()
    for _ in range(10): e.set_data(X, Y)
    alone = (time.perf_counter() - t0) / 10
    engs = [e] + [make(n) for _ in range(n - 1)]
    def work(x):
        for _ in range(10): x.set_data(X, Y)
    th = [threading.Thread(target=work, args=(x,)) for x in engs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    together = (time.perf_counter() - t0) / 10
    print(f"N={N} share 1/{n}: one update alone {1e3*alone:.2f} ms; {n} side by side: {1e3*together:.2f} ms per round = {1e3*together/n:.2f} ms per update", flush=True)
    for x in engs: x.close()
