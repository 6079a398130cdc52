"""Single-controller multi-GPU: :class:`GPEngineGroup` wraps the C-ABI's ``tgp_group_*`` (include/tgp.h).

One Python process -- so the BayesianOptimizer / Ask-Tell loop and the user's observer run once, as in the
reference (bayesian_optimizer.py:793-806) -- drives one model replica per device.  Model state is replicated
(every member runs the same deterministic ``update`` concurrently); the candidate table is sharded in contiguous
row blocks; a sharded sweep leaves each member's (value, index) winners on its device, one in-process RCCL
all-gather (or 16-byte peer copies, ``merge="peer"``) brings them to member 0, a merge kernel applies
(max value, min global index) and one small copy reaches the host.  The reference has no multi-device path
(SURVEY.md section 0); sharding and merge follow SURVEY.md section 8e.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .engine import GPEngine

_NP = np.float64


class GPEngineGroup:
    def __init__(self, d: int, kernel: str = "matern52", devices: Optional[Sequence[int]] = None,
                 merge: str = "rccl"):
        self._lib = _lib.load()
        if kernel not in _lib.KERNELS:
            raise ValueError(f"unknown kernel {kernel!r}; choose from {sorted(_lib.KERNELS)}")
        if merge not in _lib.MERGES:
            raise ValueError(f"unknown merge {merge!r}; choose from {sorted(_lib.MERGES)}")
        if devices is None:
            import torch

            devices = list(range(torch.cuda.device_count()))
        devices = [int(x) for x in devices]
        if not devices:
            raise ValueError("a group needs at least one device")
        ids = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        rc = self._lib.tgp_group_create(ids, len(devices), int(d), _lib.KERNELS[kernel], _lib.MERGES[merge], C.byref(g))
        _lib.check(self._lib, None, rc, group=True)
        self._g = g
        self.d, self.kernel, self.devices, self.merge = int(d), kernel, devices, merge
        self.N = 0
        self.M = 0
        self.members: List[GPEngine] = []
        for i, dev in enumerate(devices):
            h = C.c_void_p()
            self._chk(self._lib.tgp_group_member(self._g, i, C.byref(h)))
            self.members.append(GPEngine._borrowed(h, d, kernel, dev))

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m._h = None  # borrowed handles die with the group
            self._lib.tgp_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        _lib.check(self._lib, self._g, rc, group=True)

    def info(self) -> dict:
        n, m, r = C.c_int(), C.c_int(), C.c_int()
        self._chk(self._lib.tgp_group_info(self._g, C.byref(n), C.byref(m), C.byref(r)))
        return {"n_dev": n.value, "merge": {v: k for k, v in _lib.MERGES.items()}[m.value], "rccl_ranks": r.value}

    @property
    def primary(self) -> GPEngine:
        """Member 0: posterior queries, gradients and fits that are not sharded go here."""
        return self.members[0]

    # -- replicated model state ------------------------------------------------------------------
    def set_hyper(self, variance: float, lengthscales, noise_variance: float, mean_const: float = 0.0):
        ls = np.ascontiguousarray(np.broadcast_to(np.asarray(lengthscales, dtype=_NP), (self.d,)))
        self._chk(self._lib.tgp_group_set_hyper(self._g, float(variance), ls.ctypes.data, float(noise_variance),
                                                float(mean_const)))
        self._set_n(0)

    def _set_n(self, n: int) -> None:
        self.N = n
        for m in self.members:
            m.N = n

    def set_data(self, X, Y):
        X = np.ascontiguousarray(np.asarray(X, dtype=_NP))
        Y = np.ascontiguousarray(np.asarray(Y, dtype=_NP).reshape(-1))
        if X.ndim != 2 or X.shape[1] != self.d:
            raise ValueError(f"X must be [N, {self.d}], got {X.shape}")
        if Y.shape[0] != X.shape[0]:
            raise ValueError(f"Y must hold N={X.shape[0]} observations, got shape {Y.shape}")
        self._chk(self._lib.tgp_group_set_data(self._g, X.ctypes.data, Y.ctypes.data, X.shape[0]))
        self._set_n(X.shape[0])

    def append_data(self, Xnew, Ynew):
        X = np.ascontiguousarray(np.asarray(Xnew, dtype=_NP))
        Y = np.ascontiguousarray(np.asarray(Ynew, dtype=_NP).reshape(-1))
        if X.ndim != 2 or X.shape[1] != self.d:
            raise ValueError(f"Xnew must be [k, {self.d}], got {X.shape}")
        if Y.shape[0] != X.shape[0]:
            raise ValueError(f"Ynew must hold k={X.shape[0]} observations, got shape {Y.shape}")
        self._chk(self._lib.tgp_group_append_data(self._g, X.ctypes.data, Y.ctypes.data, X.shape[0]))
        self._set_n(self.N + X.shape[0])

    def eta(self) -> float:
        return self.primary.eta()

    # -- per-handle settings, replicated ------------------------------------------------------------------
    # A sharded sweep runs on EVERY member, so whatever changes the values a member computes (arithmetic, launch
    # policy, a penalization around pending points, the entropy tails' min-value samples) has to be set on all of
    # them -- on member 0 alone the other shards would be swept with different settings.
    def set_precision(self, precision: str = "f64") -> None:
        for m in self.members:
            m.set_precision(precision)

    def set_variant(self, v: int) -> None:
        for m in self.members:
            m.set_variant(v)

    def set_penalization(self, kind: str, pending=None, radius=None, scale=None) -> None:
        for m in self.members:
            m.set_penalization(kind, pending, radius, scale)

    def set_min_value_samples(self, samples) -> None:
        for m in self.members:
            m.set_min_value_samples(samples)

    def penalized(self, kind: str, pending, radius, scale):
        """Context manager: the penalization is active on every member inside the block only."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.set_penalization(kind, pending, radius, scale)
            try:
                yield self
            finally:
                self.set_penalization("none")

        return scope()

    # -- the sharded candidate table -----------------------------------------------------------------
    def set_candidates(self, points) -> None:
        """Scatter a host table [M, d] (e.g. ``DiscreteSearchSpace.points``) over the members."""
        P = np.ascontiguousarray(np.asarray(points, dtype=_NP))
        if P.ndim != 2 or P.shape[1] != self.d or P.shape[0] < 1:
            raise ValueError(f"candidates must be [M >= 1, {self.d}], got {P.shape}")
        self._chk(self._lib.tgp_group_set_candidates(self._g, P.ctypes.data, P.shape[0]))
        self.M = P.shape[0]

    def sample_candidates(self, seed: int, M: int, lower, upper) -> None:
        """Generate ONE logical Philox sample of M uniform candidates in the box on the devices, each member its
        own rows: the table does not depend on the number of devices."""
        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lower, dtype=_NP), (self.d,)))
        up = np.ascontiguousarray(np.broadcast_to(np.asarray(upper, dtype=_NP), (self.d,)))
        self._chk(self._lib.tgp_group_sample_candidates(self._g, int(seed), int(M), lo.ctypes.data, up.ctypes.data))
        self.M = int(M)

    # -- sharded sweeps ---------------------------------------------------------------------------
    def acq_argmax(self, acq: str, param: float):
        """Fused arg-max over the resident table -> (value, global index, point [d])."""
        bv, bi = C.c_double(), C.c_int64()
        bx = np.empty(self.d)
        self._chk(self._lib.tgp_group_acq_argmax(self._g, _lib.ACQ[acq], float(param), C.byref(bv), C.byref(bi),
                                                 bx.ctypes.data))
        return bv.value, bi.value, bx

    def acq_topk(self, acq: str, param: float, k: int):
        vals, idx = np.empty(k), np.empty(k, dtype=np.int64)
        self._chk(self._lib.tgp_group_acq_topk(self._g, _lib.ACQ[acq], float(param), int(k), vals.ctypes.data,
                                               idx.ctypes.data))
        return vals, idx

    def qei(self, Xq, eps, eta: float, jitter: float = 1e-6):
        """Batch Monte-Carlo EI of the q-batches Xq [G, q, d] (host), sharded over the members -> [G]."""
        X = np.ascontiguousarray(np.asarray(Xq, dtype=_NP))
        e = np.ascontiguousarray(np.asarray(eps, dtype=_NP))
        if X.ndim != 3 or X.shape[2] != self.d:
            raise ValueError(f"batch query points must be [G, q, {self.d}], got {X.shape}")
        G, q = X.shape[0], X.shape[1]
        if e.ndim != 2 or e.shape[0] != q:
            raise ValueError(f"eps must be [q={q}, S], got {e.shape}")
        out = np.empty(G)
        self._chk(self._lib.tgp_group_qei(self._g, X.ctypes.data, G, q, e.ctypes.data, e.shape[1], float(eta),
                                          float(jitter), out.ctypes.data))
        return out

    def trajectory(self, rff_W, rff_b, w, xi) -> "GroupTrajectory":
        return GroupTrajectory(self, rff_W, rff_b, w, xi)

    def last_kernel_ms(self) -> float:
        ms = C.c_double()
        self._chk(self._lib.tgp_group_last_kernel_ms(self._g, C.byref(ms)))
        return ms.value


class GroupTrajectory:
    """B decoupled Thompson trajectories replicated on every member of a group (tgp_group_traj_*)."""

    def __init__(self, group: GPEngineGroup, rff_W, rff_b, w, xi):
        self._group = group
        Wf = np.ascontiguousarray(rff_W, dtype=_NP)
        bf = np.ascontiguousarray(rff_b, dtype=_NP).reshape(-1)
        F = Wf.shape[0]
        if Wf.shape != (F, group.d) or bf.shape != (F,):
            raise ValueError("rff_W must be [F, d] and rff_b [F]")
        w = np.ascontiguousarray(np.asarray(w, dtype=_NP).reshape(F, -1))
        B = w.shape[1]
        xi = np.ascontiguousarray(np.asarray(xi, dtype=_NP).reshape(group.N, -1))
        if xi.shape[1] != B:
            raise ValueError(f"xi must be [N, B={B}], got {xi.shape}")
        t = C.c_void_p()
        group._chk(group._lib.tgp_group_traj_create(group._g, Wf.ctypes.data, bf.ctypes.data, F, w.ctypes.data,
                                                    xi.ctypes.data, B, C.byref(t)))
        self._t, self.F, self.B = t, F, B

    def close(self):
        if getattr(self, "_t", None):
            self._group._lib.tgp_group_traj_destroy(self._t)  # allowed after the group is gone (include/tgp.h)
            self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def argmin(self):
        """arg-min of every trajectory over the group's resident table -> (values [B], global indices [B])."""
        if not getattr(self, "_t", None) or not getattr(self._group, "_g", None):
            raise RuntimeError("the trajectory set (or its group) has been closed")
        vals, idx = np.empty(self.B), np.empty(self.B, dtype=np.int64)
        self._group._chk(self._group._lib.tgp_group_traj_argmin(self._t, vals.ctypes.data, idx.ctypes.data))
        return vals, idx
