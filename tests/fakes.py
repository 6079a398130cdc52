"""TEST INFRASTRUCTURE: a stand-in for ``trieste_amd.engine.GPEngine`` backed by the numpy oracle.

The product has no CPU path; to exercise the HOST logic (model wrapper, builders, optimizers, rules,
Ask-Tell / BO loops) without a GPU, the tests substitute this class at the engine boundary -- the
same seam the reference's tests use with their ``QuadraticMeanAndRBFKernel`` fakes
(tests/util/models/gpflow/models.py:189-210).  It mirrors GPEngine's Python surface exactly.
"""
from __future__ import annotations

import numpy as np

from oracle import gp_oracle as O


class FakeTrajectory:
    def __init__(self, eng, W, b, w, xi):
        self._eng = eng
        self.W, self.b = np.asarray(W, float), np.asarray(b, float)
        self.w = np.asarray(w, float).reshape(self.W.shape[0], -1)
        self.F, self.B = self.W.shape[0], self.w.shape[1]
        self._v = O.decoupled_weights(eng.state, self.W, self.b, self.w, np.asarray(xi, float).reshape(eng.N, -1))

    def close(self):
        pass

    def v(self):
        return self._v

    def __call__(self, Xq):
        return O.trajectory_eval(self._eng.state, self.W, self.b, self.w, self._v, np.asarray(Xq, float))

    def value_and_gradient(self, Xq):
        return O.trajectory_value_and_grad(self._eng.state, self.W, self.b, self.w, self._v, np.asarray(Xq, float))

    def argmin(self, Xq, index_base=0):
        vals = self(np.asarray(Xq, float))
        idx = np.argmin(vals, axis=0)
        return vals[idx, np.arange(self.B)], idx + index_base

    def argmin_pairs(self, Xq, index_base=0):
        return _pairs(*self.argmin(Xq, index_base))


def _pairs(vals, idx):
    """[2, V] float64: values, then the int64 indices bit-cast (the device pair layout of include/tgp.h)."""
    vals = np.atleast_1d(np.asarray(vals, np.float64))
    idx = np.atleast_1d(np.asarray(idx, np.int64))
    return np.stack([vals, idx.view(np.float64)])


class FakeRffTrajectory:
    def __init__(self, eng, W, b, eps):
        self._eng = eng
        self.W, self.b = np.asarray(W, float), np.asarray(b, float)
        self._theta = O.rff_theta(eng.state, self.W, self.b, np.asarray(eps, float))
        self.F, self.B = self._theta.shape

    def close(self):
        pass

    def theta(self):
        return self._theta

    def argmin_pairs(self, Xq, index_base=0):
        return _pairs(*self.argmin(Xq, index_base))

    def __call__(self, Xq):
        return O.rff_trajectory_eval(self._eng.state, self.W, self.b, self._theta, np.asarray(Xq, float))

    def value_and_gradient(self, Xq):
        Xq = np.asarray(Xq, float)
        zero_v = np.zeros((self._eng.N, self.B))
        scale = np.sqrt(2.0 * self._eng.state.variance / self.F)
        # the decoupled oracle with no canonical part: w = theta (its features carry the scale themselves)
        return O.trajectory_value_and_grad(self._eng.state, self.W, self.b, self._theta, zero_v, Xq)

    def argmin(self, Xq, index_base=0):
        vals = self(np.asarray(Xq, float))
        idx = np.argmin(vals, axis=0)
        return vals[idx, np.arange(self.B)], idx + index_base


class FakeEngine:
    """Same constructor and methods as GPEngine; arithmetic by oracle/gp_oracle.py."""

    created = 0
    appended = 0
    cloned = 0

    def __init__(self, d, kernel="matern52", device=0):
        if kernel not in O.KERNEL_KINDS:
            raise ValueError(f"unknown kernel {kernel!r}")
        self.d, self.kernel, self.device, self.N = int(d), kernel, device, 0
        self._hyper = None
        self.state = None
        self._pen = None
        self._ent = None
        self._rep = None
        FakeEngine.created += 1

    def close(self):
        pass

    def use_torch_stream(self):
        pass

    def use_private_stream(self):
        pass

    def set_variant(self, v):
        pass

    def set_update_concurrency(self, n=1):
        if not 1 <= int(n) <= 16:
            raise ValueError("update concurrency must be 1 ... 16")

    def set_precision(self, precision="f64"):
        if precision not in ("f64", "i8x4", "i8x5", "auto"):
            raise ValueError(f"unknown precision {precision!r}")
        self._precision = precision

    def get_precision(self):
        """tgp_get_precision: (requested, in effect, fraction the last "auto" sweep recomputed in float64 or -1).  The
        oracle-backed stand-in computes everything in float64: "auto" reports its first rung and nothing recomputed."""
        req = getattr(self, "_precision", "f64")
        if req != "auto":
            return req, req, -1.0
        return req, "i8x4", 0.0

    def release_scratch(self):
        """tgp_release_scratch: the stand-in keeps no device scratch."""

    def set_auto_sigma(self, k_sigma=8.0):
        if not (k_sigma > 0.0) or not np.isfinite(k_sigma):
            raise ValueError("k_sigma must be positive and finite")
        self._auto_sigma = float(k_sigma)

    def get_auto_report(self):
        """tgp_get_auto_report: the float64 stand-in has nothing to sample."""
        return dict(checked=0, violations=0, worst_ratio=0.0, demotions=0,
                    level=0 if getattr(self, "_precision", "f64") == "auto" else -1)

    def get_auto_strata(self):
        """tgp_get_auto_strata: the float64 stand-in has nothing to sample."""
        none = dict(checked=0, violations=0, worst_ratio=0.0)
        return dict(uniform=dict(none), adversarial=dict(none), slack_saved=0)

    def clone_from(self, other):
        if not isinstance(other, FakeEngine):
            raise TypeError(f"can only clone from an engine, got {other!r}")
        if other.d != self.d or other.kernel != self.kernel:
            raise ValueError("clone needs equal input dimension and kernel")
        self._hyper, self.state, self.N = other._hyper, other.state, other.N
        FakeEngine.cloned += 1

    def clone(self):
        twin = FakeEngine(self.d, self.kernel, self.device)
        twin.clone_from(self)
        return twin

    def set_penalization(self, kind, pending=None, radius=None, scale=None):
        if kind not in ("none", "soft", "hard"):
            raise ValueError(f"unknown penalizer {kind!r}")
        if kind == "none" or pending is None or len(pending) == 0:
            self._pen = None
            return
        pts = np.asarray(pending, float)
        if pts.ndim != 2 or pts.shape[1] != self.d:
            raise ValueError(f"pending points must be [P, {self.d}], got {pts.shape}")
        r, sc = np.asarray(radius, float).reshape(-1), np.asarray(scale, float).reshape(-1)
        if r.shape[0] != pts.shape[0] or sc.shape[0] != pts.shape[0]:
            raise ValueError("radius and scale must hold P values")
        self._pen = (kind, pts, r, sc)

    def set_min_value_samples(self, samples):
        sm = np.asarray(samples, float).reshape(-1)
        if sm.size > 4096:
            raise ValueError("number of min-value samples must be in 0..4096")
        self._ent = sm if sm.size else None

    def set_repulsion(self, twin=None, weight=1.0):
        if twin is None:
            self._rep = None
            return
        if not isinstance(twin, FakeEngine):
            raise TypeError(f"the repulsion twin must be an engine, got {twin!r}")
        if twin is self:
            raise ValueError("the repulsion twin must be a different handle")
        if twin.d != self.d or twin.kernel != self.kernel:
            raise ValueError("the repulsion twin must share input dimension, kernel and device")
        if not weight >= 0:
            raise ValueError("repulsion weight must be >= 0")
        twin._st()
        self._rep = (twin, float(weight))

    def _entropy_values(self, acq, Xq):
        if self._ent is None:
            raise RuntimeError("entropy-search acquisition needs min-value samples: call tgp_set_min_value_samples")
        m, v = self.predict(Xq)
        if acq == "mes":
            return O.min_value_entropy_search(m, v, self._ent)
        vals = O.gibbon_quality_term(m, v, self._ent, self.state.noise)
        if self._rep is not None:
            twin, w = self._rep
            _, vt = twin.predict(Xq)
            vals = vals + 0.5 * w * (np.log(vt + self.state.noise) - np.log(v + self.state.noise))
        return vals

    def penalized(self, kind, pending, radius, scale):
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.set_penalization(kind, pending, radius, scale)
            try:
                yield self
            finally:
                self.set_penalization("none")

        return scope()

    def penalization_values(self, Xq):
        if self._pen is None:
            raise RuntimeError("no penalization set: call tgp_set_penalization first")
        Xq = np.asarray(Xq, float)
        kind, pts, r, sc = self._pen
        return O.PENALIZERS[kind](Xq.reshape(-1, self.d), pts, r, sc).reshape(Xq.shape[:-1])

    def set_hyper(self, variance, lengthscales, noise_variance, mean_const=0.0):
        if not (variance > 0 and noise_variance > 0):
            raise ValueError("variance and noise_variance must be positive")
        self._hyper = (float(variance), np.broadcast_to(np.asarray(lengthscales, float), (self.d,)).copy(),
                       float(noise_variance), float(mean_const))
        self.state, self.N = None, 0

    def set_data(self, X, Y):
        if self._hyper is None:
            raise RuntimeError("tgp_set_hyper must be called before tgp_set_data")
        X = np.asarray(X, float)
        Y = np.asarray(Y, float).reshape(-1)
        if X.ndim != 2 or X.shape[1] != self.d:
            raise ValueError(f"X must be [N, {self.d}], got {X.shape}")
        if Y.shape[0] != X.shape[0]:
            raise ValueError("Y must hold N observations")
        v, ls, s2, c = self._hyper
        self._xy = (X, Y)
        try:
            self.state = O.gpr_update(self.kernel, v, ls, s2, c, X, Y)
        except np.linalg.LinAlgError as e:
            from trieste_amd._lib import NotPositiveDefiniteError

            self.state = None
            raise NotPositiveDefiniteError(str(e))
        self.N = X.shape[0]

    def append_data(self, Xnew, Ynew):
        st = self._st()
        Xnew = np.asarray(Xnew, float)
        Ynew = np.asarray(Ynew, float).reshape(-1)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.d or Ynew.shape[0] != Xnew.shape[0]:
            raise ValueError("Xnew must be [k, d] and Ynew hold k observations")
        FakeEngine.appended += 1
        self.set_data(np.concatenate([st.X, Xnew]), np.concatenate([st.Y, Ynew]))

    def _st(self):
        if self.state is None:
            raise RuntimeError("model has no data: call tgp_set_data first")
        return self.state

    def nlml(self, with_gradient=True):
        v, g = O.nlml_and_grad(self._st())
        return (v, g) if with_gradient else (v, None)

    def nlml_trial(self):
        """tgp_nlml_trial: the likelihood at the current hyper-parameters over the data of the last set_data; no
        posterior is left behind."""
        if self._hyper is None:
            raise RuntimeError("tgp_set_hyper must be called before tgp_nlml_trial")
        if getattr(self, "_xy", None) is None:
            raise RuntimeError("no data on the device: call tgp_set_data once first")
        v, ls, s2, c = self._hyper
        try:
            st = O.gpr_update(self.kernel, v, ls, s2, c, *self._xy)
        except np.linalg.LinAlgError as e:
            from trieste_amd._lib import NotPositiveDefiniteError

            raise NotPositiveDefiniteError(str(e))
        self.state, self.N = None, 0
        return O.nlml_and_grad(st)[0]

    def update_is_persistent(self, N):
        return int(N) > 3840

    def nlml_trial_batch(self, hypers):
        """tgp_nlml_trial_batch: the likelihood at every row of hypers [B, d + 3] over the data of the last set_data; the
        engine's own hyper-parameters and posterior are untouched."""
        hy = np.ascontiguousarray(hypers, dtype=np.float64)
        if hy.ndim != 2 or hy.shape[1] != self.d + 3:
            raise ValueError(f"hypers must be [B, {self.d + 3}], got {hy.shape}")
        if getattr(self, "_xy", None) is None:
            raise RuntimeError("no data on the device: call tgp_set_data once first")
        values, ok = np.full(hy.shape[0], np.nan), np.zeros(hy.shape[0], dtype=bool)
        for b, row in enumerate(hy):
            try:
                st = O.gpr_update(self.kernel, row[0], row[1:1 + self.d], row[1 + self.d], row[2 + self.d], *self._xy)
                values[b], ok[b] = O.nlml_and_grad(st)[0], True
            except np.linalg.LinAlgError:
                pass
        return values, ok

    def get_factor(self):
        st = self._st()
        W = np.linalg.inv(st.L)
        return st.L, W, W.T @ (W @ st.err)

    def predict(self, Xq):
        return O.predict(self._st(), np.asarray(Xq, float))

    def predict_mean(self, Xq):
        return O.predict(self._st(), np.asarray(Xq, float))[0]

    def predict_joint(self, Xq):
        Xq = np.asarray(Xq, float)
        if Xq.shape[-2] > 64:
            raise ValueError("q must be in 1..64")
        return O.predict_joint(self._st(), Xq)

    def sample_joint(self, Xq, eps, jitter=1e-6):
        return O.joint_samples(self._st(), np.asarray(Xq, float), np.asarray(eps, float), jitter)

    def cov_between(self, X1, X2):
        return O.covariance_between_points(self._st(), np.asarray(X1, float), np.asarray(X2, float))

    def eta(self):
        return O.eta_min_mean(self._st())

    def acq_values(self, acq, param, Xq):
        m, v = self.predict(Xq)
        if acq == "ei":
            vals = O.expected_improvement(m, v, param)
        elif acq == "pi":
            vals = O.probability_of_improvement(m, v, param)
        elif acq == "nlcb":
            vals = O.negative_lower_confidence_bound(m, v, param)
        elif acq == "aei":
            vals = O.augmented_expected_improvement(m, v, param, self.state.noise)
        elif acq in ("mes", "gibbon"):
            vals = self._entropy_values(acq, Xq)
        else:
            raise KeyError(acq)
        if self._pen is not None:
            vals = vals * self.penalization_values(Xq)
        return vals

    def acq_value_grad(self, acq, param, Xq):
        if acq in ("mes", "gibbon"):
            if self._ent is None:
                raise RuntimeError("entropy-search acquisition needs min-value samples: call tgp_set_min_value_samples")
            twin, w = self._rep if (self._rep is not None and acq == "gibbon") else (None, 0.0)
            val, grad = O.entropy_value_and_grad(self._st(), acq, self._ent, np.asarray(Xq, float),
                                                 None if twin is None else twin._st(), w)
            if self._pen is not None:  # product rule with the penalization, by finite differences of phi
                kind, pts, r, sc = self._pen
                x = np.asarray(Xq, float)
                phi = O.PENALIZERS[kind](x, pts, r, sc)
                h = 1e-6
                dphi = np.stack([(O.PENALIZERS[kind](x + h * e, pts, r, sc) - O.PENALIZERS[kind](x - h * e, pts, r, sc))
                                 / (2 * h) for e in np.eye(self.d)], axis=1)
                return val * phi, phi[:, None] * grad + val[:, None] * dphi
            return val, grad
        if self._pen is not None:
            kind, pts, r, sc = self._pen
            return O.penalized_value_and_grad(self._st(), acq, param, kind, pts, r, sc, np.asarray(Xq, float))
        return O.acq_value_and_grad(self._st(), acq, param, np.asarray(Xq, float))

    def acq_argmax(self, acq, param, Xq, index_base=0):
        Xq = np.asarray(Xq, float)
        if Xq.shape[0] == 0:
            raise ValueError("arg-max over an empty candidate set")
        vals = self.acq_values(acq, param, Xq)
        i = int(np.argmax(vals))
        return float(vals[i]), i + index_base, Xq[i].copy()

    def acq_argmax_pair(self, acq, param, Xq, index_base=0):
        v, i, _ = self.acq_argmax(acq, param, np.asarray(Xq, float), index_base)
        return _pairs(v, i)[:, 0]

    def merge_winners(self, gathered, minimize=False):
        from trieste_amd.distributed import merge_best

        g = np.asarray(gathered, np.float64)
        vals, idxs = g[:, 0, :], np.ascontiguousarray(g[:, 1, :]).view(np.int64)
        v, i = merge_best(vals, idxs, minimize)
        return _pairs(v, i)

    def synchronize(self):
        pass

    def acq_topk(self, acq, param, Xq, k, index_base=0):
        vals = self.acq_values(acq, param, np.asarray(Xq, float))
        v, i = O.top_k(vals, k)
        return v, i + index_base

    def sample_box(self, seed, first, M, lower, upper):
        from oracle.philox import sample_box

        lo = np.broadcast_to(np.asarray(lower, float), (self.d,))
        up = np.broadcast_to(np.asarray(upper, float), (self.d,))
        return sample_box(seed, first, M, lo, up)  # the engine's Philox map, bit for bit

    def qei(self, Xq, eps, eta, jitter=1e-6):
        Xq = np.asarray(Xq, float)
        lead = Xq.shape[:-2]
        out = O.batch_mc_ei(self._st(), Xq.reshape((-1,) + Xq.shape[-2:]), np.asarray(eps, float), eta, jitter)
        return out.reshape(lead)

    JOINT_SMALL_POINTS = 2048

    def joint_forward(self, Xq):
        return O.predict_joint(self._st(), np.asarray(Xq, float))

    @staticmethod
    def qei_value_grad_fits(q, S):
        return False   # the CPU tests exercise the host adjoint (joint_forward + joint_vjp), not a second oracle call

    def qei_value_grad(self, Xq, eps, eta, jitter=1e-6):
        return O.batch_mc_ei_value_and_grad(self._st(), np.asarray(Xq, float), np.asarray(eps, float), eta, jitter)

    def joint_vjp(self, Xq, gmean, gcov):
        """The engine's vector-Jacobian product of predict_joint on dense numpy arrays (K^-1 formed explicitly)."""
        st = self._st()
        Xq, gmean, gcov = np.asarray(Xq, float), np.asarray(gmean, float), np.asarray(gcov, float)
        G, q, d = Xq.shape
        ls = st.lengthscales
        Kinv = np.linalg.inv(st.L @ st.L.T)
        alpha = Kinv @ st.err
        out = np.zeros((G, q, d))
        for g in range(G):
            Gs = gcov[g] + gcov[g].T
            diff = (Xq[g][:, None, :] - st.X[None, :, :]) / ls
            r2 = np.sum(diff * diff, -1)
            Kq = O.kernel_from_r2(st.kind, st.variance, r2)                      # [q, N]
            dk = (2.0 * O._kernel_dr2(st.kind, st.variance, r2))[:, :, None] * diff / ls
            V = gmean[g][:, None] * alpha[None, :] - (Gs @ Kq) @ Kinv            # [q, N]
            out[g] = np.einsum("qnd,qn->qd", dk, V)
            dq = (Xq[g][:, None, :] - Xq[g][None, :, :]) / ls
            r2q = np.sum(dq * dq, -1)
            dkq = (2.0 * O._kernel_dr2(st.kind, st.variance, r2q))[:, :, None] * dq / ls
            off = Gs * (1.0 - np.eye(q))
            out[g] += np.einsum("ij,ijd->id", off, dkq)
        return out

    def reparam_samples(self, Xq, eps, jitter=1e-6):
        Xq = np.asarray(Xq, float)
        lead = Xq.shape[:-2]
        s = O.batch_reparam_samples(self._st(), Xq.reshape((-1,) + Xq.shape[-2:]), np.asarray(eps, float), jitter)
        return s.reshape(lead + s.shape[1:])

    def last_kernel_ms(self):
        return 0.0, 0

    def trajectory_rff(self, rff_W, rff_b, eps):
        return FakeRffTrajectory(self, rff_W, rff_b, eps)

    def trajectory(self, rff_W, rff_b, w, xi):
        return FakeTrajectory(self, rff_W, rff_b, w, xi)


class FakeGroup:
    """Stand-in for ``trieste_amd.group.GPEngineGroup``: n oracle-backed replicas, contiguous candidate shards, winners
    merged with the product's own host rule (``distributed.merge_best``)."""

    def __init__(self, d, kernel="matern52", devices=None, merge="rccl"):
        self.devices = list(devices if devices is not None else [0])
        self.members = [FakeEngine(d, kernel, dev) for dev in self.devices]
        self.d, self.kernel, self.merge, self.N, self.M = d, kernel, merge, 0, 0
        self._cand = None

    @property
    def primary(self):
        return self.members[0]

    def info(self):
        return {"n_dev": len(self.members), "merge": self.merge, "rccl_ranks": len(self.members)}

    def close(self):
        pass

    def set_hyper(self, *a, **k):
        for m in self.members:
            m.set_hyper(*a, **k)

    def set_data(self, X, Y):
        for m in self.members:
            m.set_data(X, Y)
        self.N = self.primary.N

    def append_data(self, X, Y):
        for m in self.members:
            m.append_data(X, Y)
        self.N = self.primary.N

    def eta(self):
        return self.primary.eta()

    def set_precision(self, precision="f64"):
        for m in self.members:
            m.set_precision(precision)

    def set_variant(self, v):
        for m in self.members:
            m.set_variant(v)

    def set_penalization(self, kind, pending=None, radius=None, scale=None):
        for m in self.members:
            m.set_penalization(kind, pending, radius, scale)

    def set_min_value_samples(self, samples):
        for m in self.members:
            m.set_min_value_samples(samples)

    def penalized(self, kind, pending, radius, scale):
        import contextlib

        @contextlib.contextmanager
        def scope():
            self.set_penalization(kind, pending, radius, scale)
            try:
                yield self
            finally:
                self.set_penalization("none")

        return scope()

    def set_candidates(self, points):
        self._cand = np.asarray(points, float)
        self.M = len(self._cand)

    def sample_candidates(self, seed, M, lower, upper):
        self._cand = self.primary.sample_box(seed, 0, M, lower, upper)
        self.M = M

    def _shards(self):
        from trieste_amd.distributed import shard_range

        return [shard_range(self.M, r, len(self.members)) for r in range(len(self.members))]

    def acq_argmax(self, acq, param):
        from trieste_amd.distributed import merge_best

        parts = [m.acq_argmax(acq, param, self._cand[lo:hi], index_base=lo)[:2] if hi > lo else (np.nan, -1)
                 for m, (lo, hi) in zip(self.members, self._shards())]
        v, i = merge_best(np.array([[p[0]] for p in parts]), np.array([[p[1]] for p in parts]))
        return float(v[0]), int(i[0]), self._cand[int(i[0])].copy()

    def acq_topk(self, acq, param, k):
        vals, idxs = [], []
        for m, (lo, hi) in zip(self.members, self._shards()):
            if hi > lo:
                v, i = m.acq_topk(acq, param, self._cand[lo:hi], min(k, hi - lo), index_base=lo)
                vals.append(v)
                idxs.append(i)
        vals, idxs = np.concatenate(vals), np.concatenate(idxs)
        order = np.lexsort((idxs, -vals))[:k]
        return vals[order], idxs[order]
