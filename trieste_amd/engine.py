"""Thin object wrapper over the C-ABI: one :class:`GPEngine` == one GPR model resident on one GPU.

Arrays may be numpy float64 arrays (host; staged by the library) or torch float64 CUDA tensors
(device; passed by pointer, outputs are torch tensors on the same device).  torch is plumbing
only: device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

_NP = np.float64


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class _Arg:
    """Pointer + residency of one input array (keeps the backing object alive)."""

    def __init__(self, x, shape_tail: Optional[Tuple[int, ...]] = None):
        if _is_torch(x):
            import torch

            if x.dtype != torch.float64:
                x = x.to(torch.float64)
            if x.is_cuda:
                x = x.contiguous()
                self.keep, self.ptr, self.where = x, x.data_ptr(), _lib.DEVICE
                self.shape, self.device = tuple(x.shape), x.device
                return
            x = x.numpy()
        a = np.ascontiguousarray(x, dtype=_NP)
        self.keep, self.ptr, self.where = a, a.ctypes.data, _lib.HOST
        self.shape, self.device = a.shape, None


class GPEngine:
    def __init__(self, d: int, kernel: str = "matern52", device: int = 0):
        self._lib = _lib.load()
        if kernel not in _lib.KERNELS:
            raise ValueError(f"unknown kernel {kernel!r}; choose from {sorted(_lib.KERNELS)}")
        h = C.c_void_p()
        rc = self._lib.tgp_create(int(device), int(d), _lib.KERNELS[kernel], C.byref(h))
        _lib.check(self._lib, None, rc)
        self._h = h
        self._owned = True
        self.d, self.kernel, self.device = int(d), kernel, int(device)
        self.N = 0

    @classmethod
    def _borrowed(cls, handle, d: int, kernel: str, device: int) -> "GPEngine":
        """A view of a handle owned by someone else (a :class:`~trieste_amd.group.GPEngineGroup` member):
        same methods, never destroys the handle."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h, self._owned = handle, False
        self.d, self.kernel, self.device, self.N = int(d), kernel, int(device), 0
        return self

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._lib.tgp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        _lib.check(self._lib, self._h, rc)

    def use_torch_stream(self):
        """Queue this engine's kernels on torch's current CUDA stream of its device."""
        import torch

        self._chk(self._lib.tgp_set_stream(self._h, C.c_void_p(
            torch.cuda.current_stream(self.device).cuda_stream)))
        self._private_stream = False

    def use_private_stream(self):
        """Give this engine its own (non-blocking) stream, so that several engines driven from several
        host threads overlap on the GPU -- the latency-bound factorisation chain of one model leaves most
        of the chip idle (used by ``GaussianProcessRegression.find_best_model_initialization``)."""
        self._chk(self._lib.tgp_use_private_stream(self._h))
        self._private_stream = True

    @property
    def on_private_stream(self) -> bool:
        """True after ``use_private_stream``: nothing orders this engine's kernels against torch's streams."""
        return bool(getattr(self, "_private_stream", False))

    def set_variant(self, v: int):
        self._chk(self._lib.tgp_set_variant(self._h, int(v)))

    def set_update_concurrency(self, n: int = 1):
        """``update`` on ``n`` engines at once (tgp_set_update_concurrency): this engine's persistent update kernel takes
        1 / n of the compute units, so that n engines factorising concurrently on private streams run side by side.
        The factor does not depend on n."""
        self._chk(self._lib.tgp_set_update_concurrency(self._h, int(n)))

    def set_precision(self, precision: str = "f64"):
        """Arithmetic of the plain sweeps (tgp_set_precision): "f64" (default, the parity path); "i8x4" / "i8x5" --
        W K* on the int8 matrix cores with four / five 8-bit digit planes per operand (emulated-precision throughput
        options: x5 holds the plain parity tolerance on every tested model, x4 only on well-conditioned ones);
        "auto" -- the int8 sweep with an a-posteriori repair: every candidate whose own error bound exceeds the
        parity tolerance (and, in a fused arg-max, every candidate that could still be the float64 winner) is
        recomputed in float64 inside the same call; four planes, then five, then float64 when too many candidates
        needed it.  Results hold the parity tolerance candidate by candidate and the arg-max is the float64 one -- to
        the 8-sigma error model the bounds come from (a statistical model, not a worst case); every sweep re-checks a
        uniform sample of its candidates AND the unflagged candidates whose bounds sit closest to their tolerance in float64
        against their bounds and leaves the rung when one fails (:meth:`get_auto_report`, :meth:`get_auto_strata`; the
        synchronising calls repeat their sweep on the next rung before returning, the ``*_async`` / group calls take the
        demotion at their next call).  The samples are compared, not written back: values are a pure function of
        (model, rung, candidate).  Float64 remains the only arithmetic the parity claims are made on."""
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; choose from {sorted(_lib.PRECISIONS)}")
        self._chk(self._lib.tgp_set_precision(self._h, _lib.PRECISIONS[precision]))

    def get_precision(self):
        """-> (requested, what the next plain sweep runs, fraction of its candidates the last "auto" sweep recomputed
        in float64 or -1.0): tgp_get_precision."""
        req, eff, frac = C.c_int(), C.c_int(), C.c_double()
        self._chk(self._lib.tgp_get_precision(self._h, C.byref(req), C.byref(eff), C.byref(frac)))
        names = {v: k for k, v in _lib.PRECISIONS.items()}
        return names[req.value], names[eff.value], frac.value

    def set_auto_sigma(self, k_sigma: float = 8.0):
        """K_SIGMA of the "auto" precision's per-candidate error bound (tgp_set_auto_sigma; restarts the ladder)."""
        self._chk(self._lib.tgp_set_auto_sigma(self._h, float(k_sigma)))

    def get_auto_report(self):
        """The canary of the "auto" precision since its ladder last restarted (tgp_get_auto_report) ->
        dict(checked, violations, worst_ratio, demotions, level): sampled candidates compared in float64, samples outside
        their bound, worst |var_f64 - var_int8| / bound, rungs left because of a violation, current rung (0 four planes,
        1 five, 2 float64, -1 not "auto")."""
        chk, viol, worst, dem, lev = C.c_int64(), C.c_int64(), C.c_double(), C.c_int(), C.c_int()
        self._chk(self._lib.tgp_get_auto_report(self._h, C.byref(chk), C.byref(viol), C.byref(worst), C.byref(dem),
                                                C.byref(lev)))
        return dict(checked=chk.value, violations=viol.value, worst_ratio=worst.value, demotions=dem.value, level=lev.value)

    def get_auto_strata(self):
        """The canary's report per stratum (tgp_get_auto_strata) -> dict(uniform=dict(checked, violations, worst_ratio),
        adversarial=dict(...), slack_saved): the uniform 1-in-4096 sample and the adversarial one (per 1 / 64 of a sweep the
        unflagged candidate whose bound sits closest to its tolerance)."""
        chk, viol, worst, slack = (C.c_int64 * 2)(), (C.c_int64 * 2)(), (C.c_double * 2)(), C.c_int64()
        self._chk(self._lib.tgp_get_auto_strata(self._h, chk, viol, worst, C.byref(slack)))
        return dict(uniform=dict(checked=chk[0], violations=viol[0], worst_ratio=worst[0]),
                    adversarial=dict(checked=chk[1], violations=viol[1], worst_ratio=worst[1]), slack_saved=slack.value)

    # -- model state -----------------------------------------------------------------------------
    def clone_from(self, other: "GPEngine") -> None:
        """Become a copy of ``other`` (hyper-parameters, data, cached factorisation): tgp_clone_from."""
        if not isinstance(other, GPEngine):
            raise TypeError(f"can only clone from a GPEngine, got {other!r}")
        self._chk(self._lib.tgp_clone_from(self._h, other._h))
        self.N = other.N

    def clone(self) -> "GPEngine":
        """A new engine on the same device holding a copy of this one's state."""
        twin = GPEngine(self.d, self.kernel, device=self.device)
        twin.clone_from(self)
        return twin

    def set_penalization(self, kind: str, pending=None, radius=None, scale=None) -> None:
        """Multiply every acquisition result by prod_p phi_p(x) around the pending points until cleared with
        ``kind="none"`` (tgp_set_penalization; greedy_batch.py:250-389)."""
        if kind not in _lib.PENALIZERS:
            raise ValueError(f"unknown penalizer {kind!r}; choose from {sorted(_lib.PENALIZERS)}")
        if kind == "none" or pending is None or len(pending) == 0:
            self._chk(self._lib.tgp_set_penalization(self._h, 0, None, None, None, 0))
            return
        pts = np.ascontiguousarray(np.asarray(pending, dtype=_NP))
        if pts.ndim != 2 or pts.shape[1] != self.d:
            raise ValueError(f"pending points must be [P, {self.d}], got {pts.shape}")
        P = pts.shape[0]
        r = np.ascontiguousarray(np.asarray(radius, dtype=_NP).reshape(-1))
        sc = np.ascontiguousarray(np.asarray(scale, dtype=_NP).reshape(-1))
        if r.shape[0] != P or sc.shape[0] != P:
            raise ValueError(f"radius and scale must hold P={P} values, got {r.shape}, {sc.shape}")
        self._chk(self._lib.tgp_set_penalization(self._h, _lib.PENALIZERS[kind], pts.ctypes.data, r.ctypes.data,
                                                 sc.ctypes.data, P))

    def set_min_value_samples(self, samples) -> None:
        """Samples of the objective's minimum value used by the "mes" / "gibbon" acquisition kinds
        (tgp_set_min_value_samples); an empty array clears them."""
        sm = np.ascontiguousarray(np.asarray(samples, dtype=_NP).reshape(-1))
        self._chk(self._lib.tgp_set_min_value_samples(self._h, sm.ctypes.data if sm.size else None, int(sm.size)))

    def set_repulsion(self, twin: "GPEngine" = None, weight: float = 1.0) -> None:
        """GIBBON's repulsion term: ``twin`` is this model conditioned additionally on the pending points
        (tgp_set_repulsion); None clears.  The engine keeps a reference so the twin outlives the setting."""
        if twin is None:
            self._chk(self._lib.tgp_set_repulsion(self._h, None, 0.0))
            self._twin = None
            return
        if not isinstance(twin, GPEngine):
            raise TypeError(f"the repulsion twin must be a GPEngine, got {twin!r}")
        self._chk(self._lib.tgp_set_repulsion(self._h, twin._h, float(weight)))
        self._twin = twin

    def penalization_values(self, Xq):
        """prod_p phi_p(x) of the penalization currently set, at Xq [..., d] -> [...]."""
        a, lead, M = self._flat(Xq)
        out, po = self._out(a, lead)
        self._chk(self._lib.tgp_penalization_values(self._h, a.ptr, M, po, a.where))
        return out

    def penalized(self, kind: str, pending, radius, scale):
        """Context manager: the penalization is active inside the block only."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.set_penalization(kind, pending, radius, scale)
            try:
                yield self
            finally:
                self.set_penalization("none")

        return scope()

    def set_hyper(self, variance: float, lengthscales, noise_variance: float, mean_const: float = 0.0):
        ls = np.ascontiguousarray(np.broadcast_to(np.asarray(lengthscales, dtype=_NP), (self.d,)))
        self._chk(self._lib.tgp_set_hyper(self._h, float(variance), ls.ctypes.data,
                                          float(noise_variance), float(mean_const)))
        self.N = 0

    def set_data(self, X, Y):
        ax = _Arg(X)
        ay = _Arg(Y)
        if len(ax.shape) != 2 or ax.shape[1] != self.d:
            raise ValueError(f"X must be [N, {self.d}], got {ax.shape}")
        n = ax.shape[0]
        if int(np.prod(ay.shape)) != n:
            raise ValueError(f"Y must hold N={n} observations, got shape {ay.shape}")
        if ax.where != ay.where:
            raise ValueError("X and Y must both be host arrays or both device tensors")
        self._chk(self._lib.tgp_set_data(self._h, ax.ptr, ay.ptr, n, ax.where))
        self.N = n

    def append_data(self, Xnew, Ynew):
        """Extend the data set by k rows with unchanged hyper-parameters: only the trailing block of the
        factor is recomputed (tgp_append_data)."""
        ax = _Arg(Xnew)
        ay = _Arg(Ynew)
        if len(ax.shape) != 2 or ax.shape[1] != self.d:
            raise ValueError(f"Xnew must be [k, {self.d}], got {ax.shape}")
        k = ax.shape[0]
        if int(np.prod(ay.shape)) != k:
            raise ValueError(f"Ynew must hold k={k} observations, got shape {ay.shape}")
        if ax.where != ay.where:
            raise ValueError("Xnew and Ynew must both be host arrays or both device tensors")
        self._chk(self._lib.tgp_append_data(self._h, ax.ptr, ay.ptr, k, ax.where))
        self.N += k

    def nlml(self, with_gradient: bool = True):
        """(negative log marginal likelihood, gradient [d + 3] w.r.t. lengthscales, variance, noise, mean);
        ``with_gradient=False`` returns (value, None) and skips the K^-1 product."""
        v = C.c_double()
        if not with_gradient:
            self._chk(self._lib.tgp_nlml(self._h, C.byref(v), None))
            return v.value, None
        g = np.empty(self.d + 3)
        self._chk(self._lib.tgp_nlml(self._h, C.byref(v), g.ctypes.data))
        return v.value, g

    def nlml_trial(self) -> float:
        """The value ``nlml(False)`` would give after ``set_data`` with the data already on the device, at the current
        hyper-parameters (tgp_nlml_trial: the factor only from N = 3841 on).  Leaves the engine WITHOUT a posterior --
        for the throw-away evaluations of a fit's prior draws."""
        v = C.c_double()
        self._chk(self._lib.tgp_nlml_trial(self._h, C.byref(v)))
        return v.value

    def update_is_persistent(self, N: int) -> bool:
        """Is ``set_data`` at N training points ONE persistent launch on this engine (tgp_update_is_persistent)?"""
        yes = C.c_int()
        self._chk(self._lib.tgp_update_is_persistent(self._h, int(N), C.byref(yes)))
        return bool(yes.value)

    def nlml_trial_batch(self, hypers):
        """B trial evaluations on the data already on the device (tgp_nlml_trial_batch): ``hypers`` [B, d + 3] =
        (variance, lengthscales [d], noise variance, mean) per member -> (values [B], ok [B]); a member whose kernel
        matrix is not positive definite gets NaN / False.  From N = 3841 on up to sixteen members share one persistent
        launch; the engine's own hyper-parameters and posterior are untouched."""
        hy = np.ascontiguousarray(hypers, dtype=np.float64)
        if hy.ndim != 2 or hy.shape[1] != self.d + 3:
            raise ValueError(f"hypers must be [B, {self.d + 3}] (variance, lengthscales, noise, mean), got {hy.shape}")
        B = hy.shape[0]
        values = np.empty(B, dtype=np.float64)
        status = np.zeros(B, dtype=np.int32)
        if B:
            self._chk(self._lib.tgp_nlml_trial_batch(self._h, hy.ctypes.data, B, values.ctypes.data, status.ctypes.data))
        return values, status == 0

    def release_scratch(self):
        """Free the process-wide scratch the batched trial evaluations keep on this engine's device
        (tgp_release_scratch): gigabytes at N = 4096, kept between fits on purpose; the next batched fit allocates again."""
        rc = self._lib.tgp_release_scratch(int(self.device))
        if rc != 0:
            raise RuntimeError(f"tgp_release_scratch failed with status {rc}")

    def get_factor(self):
        """(L, W = L^-1, alpha) as numpy arrays (tests / diagnostics)."""
        n = self.N
        L, W, al = np.empty((n, n)), np.empty((n, n)), np.empty(n)
        self._chk(self._lib.tgp_get_factor(self._h, L.ctypes.data, W.ctypes.data, al.ctypes.data, _lib.HOST))
        return L, W, al

    # -- outputs ---------------------------------------------------------------------------------
    @staticmethod
    def _out(arg: _Arg, shape):
        if arg.where == _lib.DEVICE:
            import torch

            t = torch.empty(shape, dtype=torch.float64, device=arg.device)
            return t, t.data_ptr()
        a = np.empty(shape, dtype=_NP)
        return a, a.ctypes.data

    def _flat(self, Xq):
        a = _Arg(Xq)
        if len(a.shape) < 1 or a.shape[-1] != self.d:
            raise ValueError(f"query points must have trailing dimension {self.d}, got {a.shape}")
        lead = a.shape[:-1]
        return a, lead, int(np.prod(lead)) if lead else 1

    def predict(self, Xq):
        a, lead, M = self._flat(Xq)
        mean, pm = self._out(a, lead)
        var, pv = self._out(a, lead)
        self._chk(self._lib.tgp_predict(self._h, a.ptr, M, pm, pv, a.where))
        return mean, var

    def predict_mean(self, Xq):
        a, lead, M = self._flat(Xq)
        mean, pm = self._out(a, lead)
        self._chk(self._lib.tgp_predict_mean(self._h, a.ptr, M, pm, a.where))
        return mean

    def predict_joint(self, Xq):
        a = _Arg(Xq)
        if len(a.shape) < 2 or a.shape[-1] != self.d:
            raise ValueError(f"joint query points must be [..., q, {self.d}], got {a.shape}")
        lead, q = a.shape[:-2], a.shape[-2]
        G = int(np.prod(lead)) if lead else 1
        mean, pm = self._out(a, lead + (q,))
        cov, pc = self._out(a, lead + (q, q))
        self._chk(self._lib.tgp_predict_joint(self._h, a.ptr, G, q, pm, pc, a.where))
        return mean, cov

    def sample_joint(self, Xq, eps, jitter: float = 1e-6):
        """Exact joint posterior samples: Xq [n, d], eps [n, S] standard normal -> [S, n]."""
        a, e = _Arg(Xq), _Arg(eps)
        if len(a.shape) != 2 or a.shape[1] != self.d:
            raise ValueError(f"query points must be [n, {self.d}], got {a.shape}")
        if len(e.shape) != 2 or e.shape[0] != a.shape[0]:
            raise ValueError(f"eps must be [n={a.shape[0]}, S], got {e.shape}")
        if a.where != e.where:
            raise ValueError("Xq and eps must both be host arrays or both be CUDA tensors")
        out, po = self._out(a, (e.shape[1], a.shape[0]))
        self._chk(self._lib.tgp_sample_joint(self._h, a.ptr, a.shape[0], e.ptr, e.shape[1], float(jitter), po, a.where))
        return out

    def cov_between(self, X1, X2):
        """X1 [P1, d], X2 [P2, d] -> posterior cross-covariance [P1, P2] (unclipped)."""
        a1, a2 = _Arg(X1), _Arg(X2)
        if len(a1.shape) != 2 or len(a2.shape) != 2 or a1.shape[1] != self.d or a2.shape[1] != self.d:
            raise ValueError(f"query points must be [P, {self.d}], got {a1.shape} and {a2.shape}")
        if a1.where != a2.where:
            raise ValueError("X1 and X2 must both be host arrays or both be CUDA tensors")
        out, po = self._out(a1, (a1.shape[0], a2.shape[0]))
        self._chk(self._lib.tgp_cov_between(self._h, a1.ptr, a1.shape[0], a2.ptr, a2.shape[0], po, a1.where))
        return out

    def eta(self) -> float:
        v = C.c_double()
        self._chk(self._lib.tgp_eta(self._h, C.byref(v)))
        return v.value

    def acq_values(self, acq: str, param: float, Xq):
        a, lead, M = self._flat(Xq)
        out, po = self._out(a, lead)
        self._chk(self._lib.tgp_acq_values(self._h, _lib.ACQ[acq], float(param), a.ptr, M, po, a.where))
        return out

    def acq_value_grad(self, acq: str, param: float, Xq):
        """Xq [P, d] -> (values [P], gradients [P, d]) of the acquisition function."""
        a, lead, P = self._flat(Xq)
        val, pv = self._out(a, lead)
        grad, pg = self._out(a, lead + (self.d,))
        self._chk(self._lib.tgp_acq_value_grad(self._h, _lib.ACQ[acq], float(param), a.ptr, P, pv, pg, a.where))
        return val, grad

    def acq_argmax(self, acq: str, param: float, Xq, index_base: int = 0):
        """-> (best value, global index, best point [d] as numpy)."""
        a, _, M = self._flat(Xq)
        bv, bi = C.c_double(), C.c_int64()
        bx = np.empty(self.d)
        self._chk(self._lib.tgp_acq_argmax(self._h, _lib.ACQ[acq], float(param), a.ptr, M, int(index_base),
                                           C.byref(bv), C.byref(bi), bx.ctypes.data, a.where))
        return bv.value, bi.value, bx

    def acq_argmax_pair(self, acq: str, param: float, Xq, index_base: int = 0):
        """The fused arg-max with the winner left ON THE DEVICE: Xq a CUDA tensor [M, d] -> CUDA float64 tensor
        [2] = (value, global index as an int64 bit pattern).  Only enqueues on the engine's stream
        (tgp_acq_argmax_async): no host synchronisation -- gather the pairs of all ranks, merge them with
        :meth:`merge_winners`, read the result once."""
        import torch

        if not _is_torch(Xq) or not Xq.is_cuda:  # host candidates: put them on this engine's device first
            Xq = torch.as_tensor(np.ascontiguousarray(Xq, dtype=_NP)).to(f"cuda:{self.device}")
        a, _, M = self._flat(Xq)
        pair = torch.empty(2, dtype=torch.float64, device=a.device)
        self._chk(self._lib.tgp_acq_argmax_async(self._h, _lib.ACQ[acq], float(param), a.ptr, M, int(index_base),
                                                 pair.data_ptr()))
        return pair

    def merge_winners(self, gathered, minimize: bool = False):
        """gathered: CUDA float64 tensor [P, 2, V] (an all-gather of P ranks' [2, V] pairs) -> CUDA tensor [2, V]
        with the winners under (max value -- min if ``minimize`` --, min global index); enqueue only."""
        import torch

        if not (_is_torch(gathered) and gathered.is_cuda and gathered.dtype == torch.float64 and gathered.dim() == 3
                and gathered.shape[1] == 2):
            raise ValueError("gathered must be a CUDA float64 tensor [P, 2, V]")
        g = gathered.contiguous()
        out = torch.empty((2, g.shape[2]), dtype=torch.float64, device=g.device)
        self._chk(self._lib.tgp_merge_winners_async(self._h, g.data_ptr(), int(g.shape[0]), int(g.shape[2]),
                                                    1 if minimize else 0, out.data_ptr()))
        return out

    def synchronize(self) -> None:
        """Wait for everything enqueued on the engine's stream (tgp_stream_synchronize)."""
        self._chk(self._lib.tgp_stream_synchronize(self._h))

    def acq_topk(self, acq: str, param: float, Xq, k: int, index_base: int = 0):
        a, _, M = self._flat(Xq)
        vals, idx = np.empty(k), np.empty(k, dtype=np.int64)
        self._chk(self._lib.tgp_acq_topk(self._h, _lib.ACQ[acq], float(param), a.ptr, M, int(index_base),
                                         int(k), vals.ctypes.data, idx.ctypes.data, a.where))
        return vals, idx

    def sample_box(self, seed: int, first: int, M: int, lower, upper):
        """Uniform candidates [M, d] generated on the device (torch CUDA tensor)."""
        import torch

        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lower, dtype=_NP), (self.d,)))
        up = np.ascontiguousarray(np.broadcast_to(np.asarray(upper, dtype=_NP), (self.d,)))
        out = torch.empty((M, self.d), dtype=torch.float64, device=f"cuda:{self.device}")
        self._chk(self._lib.tgp_sample_box(self._h, int(seed), int(first), int(M), lo.ctypes.data,
                                           up.ctypes.data, out.data_ptr()))
        return out

    def qei(self, Xq, eps, eta: float, jitter: float = 1e-6):
        a = _Arg(Xq)
        if len(a.shape) < 2 or a.shape[-1] != self.d:
            raise ValueError(f"batch query points must be [..., q, {self.d}], got {a.shape}")
        lead, q = a.shape[:-2], a.shape[-2]
        G = int(np.prod(lead)) if lead else 1
        e = _Arg(eps)
        if len(e.shape) != 2 or e.shape[0] != q:
            raise ValueError(f"eps must be [q={q}, S], got {e.shape}")
        if e.where != a.where:
            raise ValueError("Xq and eps must live in the same place (both host or both device)")
        out, po = self._out(a, lead)
        self._chk(self._lib.tgp_qei(self._h, a.ptr, G, q, e.ptr, e.shape[1], float(eta), float(jitter), po,
                                    a.where))
        return out

    JOINT_SMALL_POINTS = 2048   # tgp_joint_forward / tgp_joint_vjp: points (groups x q) per call

    def _joint_small(self, Xq):
        a = _Arg(Xq)
        if len(a.shape) != 3 or a.shape[-1] != self.d:
            raise ValueError(f"batch query points must be [G, q, {self.d}], got {a.shape}")
        G, q = a.shape[0], a.shape[1]
        if G * q > self.JOINT_SMALL_POINTS:
            raise ValueError(f"{G} groups of {q} points: at most {self.JOINT_SMALL_POINTS} points per call (chunk the groups)")
        return a, G, q

    def joint_forward(self, Xq):
        """Xq [G, q, d] (G * q <= 2048) -> mean [G, q], cov [G, q, q]: ``predict_joint`` of the handful of q-batches an
        L-BFGS-B iteration holds, as a skinny product (tgp_joint_forward)."""
        a, G, q = self._joint_small(Xq)
        mean, pm = self._out(a, (G, q))
        cov, pc = self._out(a, (G, q, q))
        if G:
            self._chk(self._lib.tgp_joint_forward(self._h, a.ptr, G, q, pm, pc, a.where))
        return mean, cov

    def joint_vjp(self, Xq, gmean, gcov):
        """d/dXq [G, q, d] of sum gmean * mean + sum gcov * cov (tgp_joint_vjp): the engine's share of a batch acquisition
        function's gradient."""
        a, G, q = self._joint_small(Xq)
        gm, gc = _Arg(gmean), _Arg(gcov)
        if gm.shape != (G, q) or gc.shape != (G, q, q):
            raise ValueError(f"gmean must be [{G}, {q}] and gcov [{G}, {q}, {q}], got {gm.shape} and {gc.shape}")
        if gm.where != a.where or gc.where != a.where:
            raise ValueError("Xq, gmean and gcov must live in the same place (all host or all device)")
        grad, pg = self._out(a, (G, q, self.d))
        if G:
            self._chk(self._lib.tgp_joint_vjp(self._h, a.ptr, G, q, gm.ptr, gc.ptr, pg, a.where))
        return grad

    @staticmethod
    def qei_value_grad_fits(q: int, S: int) -> bool:
        """Does tgp_qei_value_grad take groups of q points with S draws (one wave per group, everything in LDS)?"""
        return 1 <= q <= 64 and 8 * (2 * q * (q | 1) + 128) + 4 * ((S + 1) & ~1) <= 160 * 1024

    def qei_value_grad(self, Xq, eps, eta: float, jitter: float = 1e-6):
        """Xq [G, q, d] (G * q <= 2048, q <= 64), eps [q, S] -> (qEI [G], gradient [G, q, d]) in one device call
        (tgp_qei_value_grad)."""
        a, G, q = self._joint_small(Xq)
        e = _Arg(eps)
        if len(e.shape) != 2 or e.shape[0] != q:
            raise ValueError(f"eps must be [q={q}, S], got {e.shape}")
        if e.where != a.where:
            raise ValueError("Xq and eps must live in the same place (both host or both device)")
        val, pv = self._out(a, (G,))
        grad, pg = self._out(a, (G, q, self.d))
        if G:
            self._chk(self._lib.tgp_qei_value_grad(self._h, a.ptr, G, q, e.ptr, e.shape[1], float(eta), float(jitter), pv, pg,
                                                   a.where))
        return val, grad

    def reparam_samples(self, Xq, eps, jitter: float = 1e-6):
        """Xq [..., q, d], eps [q, S] -> samples [..., S, q]."""
        a = _Arg(Xq)
        if len(a.shape) < 2 or a.shape[-1] != self.d:
            raise ValueError(f"batch query points must be [..., q, {self.d}], got {a.shape}")
        lead, q = a.shape[:-2], a.shape[-2]
        G = int(np.prod(lead)) if lead else 1
        e = _Arg(eps)
        if len(e.shape) != 2 or e.shape[0] != q:
            raise ValueError(f"eps must be [q={q}, S], got {e.shape}")
        if e.where != a.where:
            raise ValueError("Xq and eps must live in the same place (both host or both device)")
        S = e.shape[1]
        out, po = self._out(a, lead + (S, q))
        self._chk(self._lib.tgp_reparam_samples(self._h, a.ptr, G, q, e.ptr, S, float(jitter), po, a.where))
        return out

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_int()
        self._chk(self._lib.tgp_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- trajectories ----------------------------------------------------------------------------
    def trajectory_rff(self, rff_W, rff_b, eps) -> "Trajectory":
        """B trajectories f_b(x) = phi(x) . theta_b + c with theta ~ the posterior of the RFF weights given
        the data (reference RandomFourierFeatureTrajectorySampler), theta = mean + chol(cov) eps[:, b]."""
        return Trajectory(self, rff_W, rff_b, eps, None)

    def trajectory(self, rff_W, rff_b, w, xi) -> "Trajectory":
        return Trajectory(self, rff_W, rff_b, w, xi)


class Trajectory:
    """B decoupled trajectories sharing one RFF basis (tgp_traj_*)."""

    def __init__(self, eng: GPEngine, rff_W, rff_b, w, xi=None):
        """Decoupled trajectories from prior weights w [F, B] and noise draws xi [N, B]; with
        ``xi=None`` ``w`` holds standard-normal draws eps [F, B] for the RFF weight posterior
        (tgp_traj_create_rff: no canonical part)."""
        self._eng = eng
        Wf = np.ascontiguousarray(rff_W, dtype=_NP)
        bf = np.ascontiguousarray(rff_b, dtype=_NP).reshape(-1)
        F = Wf.shape[0]
        if Wf.shape != (F, eng.d) or bf.shape != (F,):
            raise ValueError("rff_W must be [F, d] and rff_b [F]")
        w = np.ascontiguousarray(np.asarray(w, dtype=_NP).reshape(F, -1))
        B = w.shape[1]
        t = C.c_void_p()
        if xi is None:
            rc = eng._lib.tgp_traj_create_rff(eng._h, Wf.ctypes.data, bf.ctypes.data, F, w.ctypes.data, B, C.byref(t))
        else:
            xi = np.ascontiguousarray(np.asarray(xi, dtype=_NP).reshape(eng.N, -1))
            if xi.shape[1] != B:
                raise ValueError(f"xi must be [N, B={B}], got {xi.shape}")
            rc = eng._lib.tgp_traj_create(eng._h, Wf.ctypes.data, bf.ctypes.data, F, w.ctypes.data,
                                          xi.ctypes.data, B, C.byref(t))
        eng._chk(rc)
        self._t, self.F, self.B = t, F, B

    def theta(self):
        """Feature weights [F, B] of an RFF-weight trajectory."""
        self._live()
        out = np.empty((self.F, self.B))
        self._eng._chk(self._eng._lib.tgp_traj_get_theta(self._t, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_t", None):
            self._eng._lib.tgp_traj_destroy(self._t)
            self._t = None

    def _live(self):
        """The C trajectory keeps a pointer to its model handle: refuse to run once either is closed."""
        if not getattr(self, "_t", None) or not getattr(self._eng, "_h", None):
            raise RuntimeError("the trajectory (or the engine it belongs to) has been closed")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def v(self):
        self._live()
        out = np.empty((self._eng.N, self.B))
        self._eng._chk(self._eng._lib.tgp_traj_get_v(self._t, out.ctypes.data))
        return out

    def __call__(self, Xq):
        """Xq [M, d] (shared) or [M, B, d] (per-trajectory inputs) -> [M, B]."""
        self._live()
        a = _Arg(Xq)
        d = self._eng.d
        if len(a.shape) == 2 and a.shape[1] == d:
            per, M = 0, a.shape[0]
        elif len(a.shape) == 3 and a.shape[1] == self.B and a.shape[2] == d:
            per, M = 1, a.shape[0]
        else:
            raise ValueError(f"trajectory inputs must be [M, {d}] or [M, {self.B}, {d}], got {a.shape}")
        out, po = GPEngine._out(a, (M, self.B))
        self._eng._chk(self._eng._lib.tgp_traj_eval(self._t, a.ptr, M, per, po, a.where))
        return out

    def value_and_gradient(self, Xq):
        """Xq [P, B, d] (trajectory b at its own point) -> (values [P, B], gradients [P, B, d])."""
        self._live()
        a = _Arg(Xq)
        d = self._eng.d
        if len(a.shape) != 3 or a.shape[1] != self.B or a.shape[2] != d:
            raise ValueError(f"trajectory inputs must be [P, {self.B}, {d}], got {a.shape}")
        P = a.shape[0]
        val, pv = GPEngine._out(a, (P, self.B))
        grad, pg = GPEngine._out(a, (P, self.B, d))
        self._eng._chk(self._eng._lib.tgp_traj_value_grad(self._t, a.ptr, P, pv, pg, a.where))
        return val, grad

    def argmin_pairs(self, Xq, index_base: int = 0):
        """The arg-min of all B trajectories with the winners left ON THE DEVICE: Xq a CUDA tensor [M, d] -> CUDA
        float64 tensor [2, B] (values, then global indices as int64 bit patterns); enqueue only
        (tgp_traj_argmin_async)."""
        import torch

        self._live()
        a = _Arg(Xq)
        if len(a.shape) != 2 or a.shape[1] != self._eng.d or a.where != _lib.DEVICE:
            raise ValueError("argmin_pairs needs device-resident candidates [M, d] (a CUDA tensor)")
        pairs = torch.empty((2, self.B), dtype=torch.float64, device=a.device)
        self._eng._chk(self._eng._lib.tgp_traj_argmin_async(self._t, a.ptr, a.shape[0], int(index_base),
                                                            pairs.data_ptr()))
        return pairs

    def argmin(self, Xq, index_base: int = 0):
        self._live()
        a = _Arg(Xq)
        if len(a.shape) != 2 or a.shape[1] != self._eng.d:
            raise ValueError("arg-min candidates must be [M, d]")
        vals, idx = np.empty(self.B), np.empty(self.B, dtype=np.int64)
        self._eng._chk(self._eng._lib.tgp_traj_argmin(self._t, a.ptr, a.shape[0], int(index_base),
                                                      vals.ctypes.data, idx.ctypes.data, a.where))
        return vals, idx
