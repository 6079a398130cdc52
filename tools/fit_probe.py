import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from trieste_amd import objectives as O
w = dict(bench.WORKLOADS["headline"])
X, Y = O.synthetic_problem(getattr(O, w["objective"]), w["d"], w["N"])
for i in range(3):
    print(i, bench.end_to_end_fit_ms(X, Y, w), flush=True)
print('acquire', bench.end_to_end_acquire_ms(X, Y, w))
for i in range(2):
    print(i, bench.end_to_end_fit_ms(X, Y, w), flush=True)
