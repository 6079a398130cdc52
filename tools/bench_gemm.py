"""Rate of the factor GEMM kernels in isolation (development aid; uses the unexported-from-header tgp_debug_gemm)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd.engine import GPEngine
eng = GPEngine(2, "rbf")
lib = eng._lib
lib.tgp_debug_gemm.restype = C.c_int
lib.tgp_debug_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
def run(m, n, k, tb, tri, lo=0, reps=5):
    ms = C.c_double()
    rc = lib.tgp_debug_gemm(eng._h, m, n, k, tb, tri, lo, reps, C.byref(ms))
    assert rc == 0, rc
    frac = {0: 1.0, 1: 0.5, 2: 0.5, 3: 0.5, 4: 1/3., 5: 0.5}[tri] * (0.5 if lo else 1.0)
    fl = 2.0 * m * n * k * frac
    print(f"m={m} n={n} k={k} tb={tb} tri={tri} lower={lo}: {ms.value*1e3:9.1f} us  {fl/ms.value*1e-9:6.1f} TFLOP/s (effective, pruned flops)", flush=True)
for s in [int(a) for a in sys.argv[1:]] or (512, 1024, 2048, 4096, 8192):
    for tb, tri, lo in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 3, 0), (1, 0, 1)):
        run(s, s, s, tb, tri, lo)
