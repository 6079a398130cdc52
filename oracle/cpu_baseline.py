"""CPU baselines for bench.py's ``cpu_baseline`` leg.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

The reference's own path (GPflow / TensorFlow on the host cores) cannot be installed here,
so the baseline is a restatement of ITS algorithmic shape on the same class of kernels TensorFlow/Eigen dispatch to
(MKL ``trsm`` / ``gemm`` through torch-CPU, float64), as SURVEY.md section 8(d) specifies:

* ``reference_shape``: per chunk of ``chunk`` candidates materialise K* [N, chunk] (gpflow kernel:
  scaled square distance in the |a|^2 + |b|^2 - 2ab form), A = L^-1 K* and L^-T A (the TWO triangular solves of
  gpflow's ``base_conditional_with_lm`` behind ``predict_f``, reference models/gpflow/interface.py:119-124), column
  norms, clip, EI (acquisition/function/function.py:220-223), running arg-max (acquisition/optimizer.py:149-150);
  the chunking mirrors ``split_acquisition_function_calls`` (acquisition/utils.py:31-109).
* ``improved``: the same result with what the engine's algorithm adds -- cached alpha = K^-1 (Y - c) (mean = K*^T
  alpha, one ``gemv``) and ONE solve for the variance -- so that speed-ups are not inflated by the algorithmic change.

Both are checked against oracle.gp_oracle in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np


def _kernel_from_r2(kind: str, variance: float, r2):
    import torch

    if kind == "rbf":
        return variance * torch.exp(-0.5 * r2)
    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    if kind == "matern12":
        return variance * torch.exp(-r)
    if kind == "matern32":
        s = math.sqrt(3.0) * r
        return variance * (1.0 + s) * torch.exp(-s)
    s = math.sqrt(5.0) * r
    return variance * (1.0 + s + 5.0 / 3.0 * torch.clamp(r2, min=1e-36)) * torch.exp(-s)


class TorchCpuSweep:
    """Holds the model state as torch CPU float64 tensors (taken from an oracle.gp_oracle.GPRState)."""

    def __init__(self, state, threads: int | None = None):
        import torch

        self.torch = torch
        self.threads = int(threads or os.cpu_count() or 1)
        torch.set_num_threads(self.threads)
        self.kind, self.variance, self.c = state.kind, float(state.variance), float(state.mean_const)
        self.Xs = torch.from_numpy(np.ascontiguousarray(state.X / state.lengthscales))
        self.ls = torch.from_numpy(np.ascontiguousarray(state.lengthscales))
        self.xn = (self.Xs * self.Xs).sum(1)
        self.L = torch.from_numpy(np.ascontiguousarray(state.L))
        self.err = torch.from_numpy(np.ascontiguousarray(state.err))[:, None]
        self.alpha = torch.cholesky_solve(self.err, self.L)  # cached once per update ("improved" only)

    def _kstar(self, Xq):
        q = Xq / self.ls
        r2 = self.xn[:, None] + (q * q).sum(1)[None, :] - 2.0 * (self.Xs @ q.T)
        return _kernel_from_r2(self.kind, self.variance, torch_clamp0(self.torch, r2))

    def _ei(self, mean, var, eta):
        torch = self.torch
        sd = torch.sqrt(var)
        z = (eta - mean) / sd
        cdf = 0.5 * torch.erfc(-z / math.sqrt(2.0))
        pdf = torch.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
        return (eta - mean) * cdf + sd * pdf

    def chunk_mean_var(self, Xq_np, improved: bool = False):
        """(mean [chunk], var [chunk]) of one chunk, reference interface.py:119-124 (clip at 1e-12 included)."""
        torch = self.torch
        Xq = torch.from_numpy(np.ascontiguousarray(Xq_np))
        Ks = self._kstar(Xq)                                                        # [N, chunk]
        A = torch.linalg.solve_triangular(self.L, Ks, upper=False)                  # L^-1 K*
        var = torch.clamp(self.variance - (A * A).sum(0), min=1e-12)
        if improved:
            mean = (Ks.T @ self.alpha)[:, 0] + self.c
        else:
            A2 = torch.linalg.solve_triangular(self.L.T, A, upper=True)             # L^-T A
            mean = (A2.T @ self.err)[:, 0] + self.c
        return mean, var

    def chunk_values(self, Xq_np, eta: float, improved: bool):
        mean, var = self.chunk_mean_var(Xq_np, improved)
        return self._ei(mean, var, eta)

    def sweep(self, Xq_np, eta: float, chunk: int, improved: bool = False):
        best, best_i = -np.inf, -1
        for s in range(0, Xq_np.shape[0], chunk):
            vals = self.chunk_values(Xq_np[s:s + chunk], eta, improved)
            i = int(self.torch.argmax(vals))
            v = float(vals[i])
            if v > best:
                best, best_i = v, s + i
        return best, best_i


def torch_clamp0(torch, r2):
    return torch.clamp(r2, min=0.0)


def best_thread_count(state, eta: float, d: int, improved: bool, probe: int = 2048) -> int:
    """MKL's triangular solves stop scaling (and regress) well before 256 hyper-threads: time one small chunk at
    every power-of-two fraction of the logical cores down to 16 and keep the fastest, so the baseline is the best
    this host can do rather than whatever the default thread count gives."""
    import torch

    ncpu = os.cpu_count() or 1
    cands = sorted({max(1, ncpu >> s) for s in range(0, 5)} | {min(ncpu, 16)}, reverse=True)
    X = np.random.default_rng(1).uniform(size=(probe, d))
    best_t, best_n = None, cands[0]
    for n in cands:
        sw = TorchCpuSweep(state, threads=n)
        sw.chunk_values(X[:256], eta, improved)
        t0 = time.perf_counter()
        sw.chunk_values(X, eta, improved)
        t = time.perf_counter() - t0
        if best_t is None or t < best_t:
            best_t, best_n = t, n
    torch.set_num_threads(best_n)
    return best_n


def timed_sweep(state, eta: float, d: int, chunk: int, budget_s: float, improved: bool, seed: int = 5678,
                max_chunks: int = 64, threads: int | None = None):
    """Sweep fresh uniform chunks for about ``budget_s`` seconds -> (candidates / s, candidates done, seconds, threads)."""
    sw = TorchCpuSweep(state, threads=threads or best_thread_count(state, eta, d, improved))
    rng = np.random.default_rng(seed)
    sw.chunk_values(rng.uniform(size=(min(chunk, 2048), d)), eta, improved)  # warm-up (thread pools, allocations)
    done, t0 = 0, time.perf_counter()
    while True:
        sw.sweep(rng.uniform(size=(chunk, d)), eta, chunk, improved)
        done += chunk
        el = time.perf_counter() - t0
        if el > budget_s or done >= max_chunks * chunk:
            break
    return done / el, done, el, sw.threads
