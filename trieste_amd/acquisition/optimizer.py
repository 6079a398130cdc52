"""Acquisition optimizers of the hot path: the candidate sweep + arg-max
(reference trieste/acquisition/optimizer.py: automatic_optimizer_selector 90-121,
_get_max_discrete_points 124-150, optimize_discrete 153-170, generate_initial_points 247-341,
batchify_joint / batchify_vectorize 897-970, generate_random_search_optimizer 973-1011).

An ``AcquisitionOptimizer`` is ``(search_space, acquisition_function | (function, V)) -> points [V or 1, D]``
and *maximises*.  When the function is one of this package's engine-backed objects (it exposes
``argmax`` / ``top_k``) the sweep runs fused on the GPU: predict + acquisition + arg-max without
the [M] values ever being written out.  Any other callable takes the generic path (evaluate, then
arg-max on the values it returned).

Gradient refinement of the sweep winners (``generate_continuous_optimizer``, optimizer.py:344-745) is
here too: the best ``num_optimization_runs`` points of the initial sweep start as many L-BFGS-B runs
(scipy), each in its own greenlet so that the value-and-gradient evaluations of one iteration of ALL
runs go to the GPU as a single batch (``tgp_acq_value_grad``); the reference obtains the gradient by
TensorFlow autodiff (optimizer.py:628-629), the engine computes the same derivative analytically.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterator, Optional, Tuple

import numpy as np
import scipy.optimize as spo

from ..space import Box, DiscreteSearchSpace, SearchSpace

NUM_SAMPLES_MIN = 5000   # optimizer.py:46-66
NUM_SAMPLES_DIM = 1000
NUM_RUNS_DIM = 10


class FailedOptimizationError(Exception):
    """Raised when an acquisition optimizer fails (optimizer.py:69-70)."""


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _fresh_seed() -> int:
    """A seed for the device-side candidate generator drawn from the package's host generator (reproducible under
    ``trieste_amd.set_seed``, fresh otherwise)."""
    from ..rng import make_rng

    return int(make_rng().integers(0, 2 ** 31 - 1))


def _split(target_func) -> Tuple[Callable, int]:
    if isinstance(target_func, tuple):
        fn, V = target_func
    else:
        fn, V = target_func, 1
    if V <= 0:
        raise ValueError(f"vectorization must be positive, got {V}")
    return fn, V


def _to_host(x) -> np.ndarray:
    return x.cpu().numpy() if _is_torch(x) else np.asarray(x)


def _get_max_discrete_points(points, target_func) -> np.ndarray:
    """points [M, 1, D] -> the V maximisers [V, D]; first index wins ties (tf.math.argmax)."""
    fn, V = _split(target_func)
    if V == 1 and hasattr(fn, "argmax"):  # fused device sweep
        _, _, x = fn.argmax(points[:, 0, :])
        return np.asarray(x)[None, :]
    if _is_torch(points):
        tiled = points.repeat(1, V, 1)
    else:
        tiled = np.tile(points, [1, V, 1])
    values = _to_host(fn(tiled))
    if values.ndim != 2 or values.shape[1] != V:
        raise ValueError(f"The result of function target_func has shape {values.shape}, however, expected a "
                         f"trailing dimension of size {V}.")
    best = np.argmax(values, axis=0)  # [V], first index on ties
    host_pts = _to_host(tiled)
    return host_pts[best, np.arange(V), :]


def optimize_discrete(space: DiscreteSearchSpace, target_func) -> np.ndarray:
    """Evaluate every point of a discrete space (optimizer.py:153-170)."""
    return _get_max_discrete_points(space.points[:, None, :], target_func)


def generate_random_search_optimizer(num_samples: int = NUM_SAMPLES_MIN, seed: Optional[int] = None,
                                     on_device: bool = True):
    """Random-search optimizer over ``num_samples`` points of the space (optimizer.py:973-1011).
    With an engine-backed acquisition function and a Box the candidates are generated on the GPU
    (Philox) so a 10^6..10^8-candidate sweep never touches the host."""
    if num_samples <= 0:
        raise ValueError(f"num_samples must be positive, got {num_samples}")

    def optimize_random(space: SearchSpace, target_func) -> np.ndarray:
        fn, V = _split(target_func)
        eng = getattr(fn, "_engine", None)
        if on_device and V == 1 and eng is not None and hasattr(fn, "argmax") and isinstance(space, Box) \
                and hasattr(eng, "sample_box"):
            # unseeded: a fresh candidate set per call, like space.sample(n) (not the same Philox stream every step)
            the_seed = _fresh_seed() if seed is None else seed
            if hasattr(fn, "argmax_sampled"):  # candidates generated where they are swept (all GPUs of a group)
                _, _, x = fn.argmax_sampled(the_seed, num_samples, space.lower, space.upper)
            else:
                _, _, x = fn.argmax(space.sample_device(eng, num_samples, seed=the_seed))
            return np.asarray(x)[None, :]
        points = space.sample(num_samples, seed=seed)[:, None, :]
        return _get_max_discrete_points(points, target_func)

    return optimize_random


def automatic_optimizer_selector(space: SearchSpace, target_func) -> np.ndarray:
    """Pick an optimizer for the space (optimizer.py:90-121): exhaustive for discrete spaces; for a
    Box the continuous optimizer with max(5000, 1000 * D) initial samples and 10 * D L-BFGS-B runs.
    Functions that expose no gradient (foreign callables) are swept by random search only; a batch function with
    ``value_and_gradient`` (qEI since round 6) reaches this selector flattened by ``batchify_joint`` and is refined too."""
    if isinstance(space, DiscreteSearchSpace):
        return optimize_discrete(space, target_func)
    if isinstance(space, Box):
        num_samples = max(NUM_SAMPLES_MIN, NUM_SAMPLES_DIM * space.dimension)
        fn, V = _split(target_func)
        if hasattr(fn, "value_and_gradient"):
            return generate_continuous_optimizer(num_initial_samples=num_samples,
                                                 num_optimization_runs=NUM_RUNS_DIM * space.dimension)(space, target_func)
        return generate_random_search_optimizer(num_samples)(space, target_func)
    raise NotImplementedError(f"No optimizer currently supports acquisition function maximisation over search "
                              f"spaces of type {space}. Try specifying the optimize_random optimizer")


def generate_initial_points(num_initial_points: int, initial_sampler, space: SearchSpace, target_func,
                            vectorization: int = 1) -> np.ndarray:
    """Best ``num_initial_points`` of the candidates produced by ``initial_sampler(space)`` (an
    iterable of [M_b, D] batches): running top-k, descending, ties by lower index
    (optimizer.py:247-341).  Returns [k, V, D]."""
    top_vals = None  # [V, k]
    top_pts = None   # [V, k, D]
    for candidates in initial_sampler(space):
        cand = _to_host(candidates)
        if cand.ndim != 2:
            raise ValueError(f"The initial samples must be a tensor of rank 2, got a tensor of rank {cand.ndim}.")
        if vectorization == 1 and hasattr(target_func, "top_k"):
            k = min(num_initial_points, cand.shape[0])
            v, i = target_func.top_k(candidates, k)
            vals, pts = np.asarray(v)[None, :], cand[np.asarray(i)][None, :, :]
        else:
            tiled = np.tile(cand[:, None, :], [1, vectorization, 1])
            values = _to_host(target_func(tiled))
            if values.ndim != 2 or values.shape[1] != vectorization:
                raise ValueError(f"The result of function target_func has shape {values.shape}, however, expected "
                                 f"a trailing dimension of size {vectorization}.")
            vals, pts = values.T, np.transpose(tiled, [1, 0, 2])
        top_vals = vals if top_vals is None else np.concatenate([top_vals, vals], axis=1)
        top_pts = pts if top_pts is None else np.concatenate([top_pts, pts], axis=1)
        k = min(num_initial_points, top_vals.shape[1])
        order = np.stack([np.lexsort((np.arange(top_vals.shape[1]), -top_vals[v]))[:k]
                          for v in range(top_vals.shape[0])])
        top_vals = np.take_along_axis(top_vals, order, axis=1)
        top_pts = np.take_along_axis(top_pts, order[:, :, None], axis=1)
    if top_pts is None:
        raise ValueError("No initial point generated!")
    return np.transpose(top_pts, [1, 0, 2])


def sample_from_space(num_samples: int, batch_size: Optional[int] = None, seed: Optional[int] = None):
    """An initial-point sampler yielding ``num_samples`` points of the space in batches
    (optimizer.py:196-244)."""
    if num_samples <= 0:
        raise ValueError(f"num_samples must be positive, got {num_samples}")
    if batch_size is not None and batch_size <= 0:
        raise ValueError(f"batch_size must be positive, got {batch_size}")
    bs = batch_size or num_samples

    def sampler(space: SearchSpace) -> Iterator[np.ndarray]:
        done, it = 0, 0
        while done < num_samples:
            n = min(bs, num_samples - done)
            yield space.sample(n, seed=None if seed is None else seed + it)
            done += n
            it += 1

    return sampler


# L-BFGS-B from many starts with ONE driver loop around scipy's own core routine (scipy.optimize._lbfgsb.setulb, the routine
# scipy.optimize.minimize(method="L-BFGS-B") itself steps): the same iterates, values and evaluation counts as a
# scipy.optimize.minimize call per start -- tests/test_host_logic.py holds the two against each other -- without the per-start
# ScalarFunction / OptimizeResult / greenlet machinery, which is most of a default acquire's time (80 starts at d = 8: ~9 of
# 12 ms at N = 4096, tools/prof_acquire.py).  Used when the optimizer arguments are the ones it understands and the private
# routine has the signature of the scipy this was written against; the greenlet form below remains the fallback.
LOCKSTEP_LBFGSB = True
_LOCKSTEP_OPTIONS = {"maxcor": 10, "ftol": 2.2204460492503131e-09, "gtol": 1e-5, "maxfun": 15000, "maxiter": 15000, "maxls": 20}


def _lockstep_options(args: Dict[str, Any]):
    """The L-BFGS-B options of ``optimizer_args`` if the lock-step driver understands all of them, else None."""
    if not LOCKSTEP_LBFGSB or set(args) - {"options"}:
        return None
    opts = dict(_LOCKSTEP_OPTIONS)
    given = args.get("options") or {}
    if set(given) - set(opts):
        return None
    opts.update(given)
    if not opts["maxls"] > 0:
        return None
    try:
        from scipy.optimize import _lbfgsb
        if not hasattr(_lbfgsb, "setulb") or tuple(int(v) for v in __import__("scipy").__version__.split(".")[:2]) != (1, 15):
            return None
    except Exception:
        return None
    return opts


def _lockstep_lbfgsb(evaluate, flat: np.ndarray, lower: np.ndarray, upper: np.ndarray, opts: Dict[str, Any]):
    """Minimise from every row of ``flat`` [R, D]; ``evaluate(x [P, D], rows [P]) -> (f [P], g [P, D])`` is called with ALL
    runs that are waiting for a value.  -> (success [R], f [R], x [R, D], nfev [R]) as scipy.optimize.minimize reports them
    (scipy/optimize/_lbfgsb_py.py:_minimize_lbfgsb is the loop this restates, one run at a time there)."""
    from scipy.optimize import _lbfgsb

    R, n = flat.shape
    m, maxls = int(opts["maxcor"]), int(opts["maxls"])
    factr, pgtol = opts["ftol"] / np.finfo(float).eps, opts["gtol"]
    nbd = np.zeros(n, np.int32)
    low, up = np.zeros(n, np.float64), np.zeros(n, np.float64)
    for i in range(n):  # (the bound codes of L-BFGS-B: 0 none, 1 lower, 2 both, 3 upper)
        has_l, has_u = not np.isinf(lower[i]), not np.isinf(upper[i])
        if has_l:
            low[i] = lower[i]
        if has_u:
            up[i] = upper[i]
        nbd[i] = (2 if has_u else 1) if has_l else (3 if has_u else 0)
    x = np.clip(np.array(flat, dtype=np.float64), lower, upper)
    f = [0.0] * R
    g = np.zeros((R, n), np.float64)
    wa = np.zeros((R, 2 * m * n + 5 * n + 11 * m * m + 8 * m), np.float64)
    iwa = np.zeros((R, 3 * n), np.int32)
    task, ln_task = np.zeros((R, 2), np.int32), np.zeros((R, 2), np.int32)
    lsave, isave, dsave = np.zeros((R, 4), np.int32), np.zeros((R, 44), np.int32), np.zeros((R, 29), np.float64)
    nit, nfev = np.zeros(R, np.int64), np.zeros(R, np.int64)
    active = list(range(R))
    while active:
        need, still = [], []
        for i in active:
            while True:
                _lbfgsb.setulb(m, x[i], low, up, nbd, f[i], g[i], factr, pgtol, wa[i], iwa[i], task[i], lsave[i], isave[i],
                               dsave[i], maxls, ln_task[i])
                if task[i, 0] == 3:     # value and gradient at x[i], please
                    need.append(i)
                    still.append(i)
                    break
                if task[i, 0] != 1:     # converged, stopped or failed
                    break
                nit[i] += 1             # a new iterate: the limits (an excess of evaluations is noticed here, as in scipy)
                if nit[i] >= opts["maxiter"]:
                    task[i] = (5, 504)
                elif nfev[i] > opts["maxfun"]:
                    task[i] = (5, 502)
        active = still
        if need:
            rows = np.array(need)
            vals, grads = evaluate(x[rows], rows)
            for j, i in enumerate(need):
                f[i] = float(vals[j])
                g[i] = grads[j]
                nfev[i] += 1
    return task[:, 0] == 4, np.array(f, dtype=np.float64), x, nfev


def _perform_parallel_continuous_optimization(fn, space: Box, starting_points: np.ndarray,
                                              optimizer_args: Dict[str, Any]):
    """L-BFGS-B from every start at once (optimizer.py:563-698): each run lives in a greenlet that
    hands its current iterate to the parent; the parent evaluates value and gradient of ALL pending
    iterates in one device call and resumes the runs.  Maximises ``fn`` (scipy minimises its
    negation).  ``starting_points`` [R, D] for a batch-size-one function (``fn.value_and_gradient``
    maps [P, D] -> ([P], [P, D])) or [R, V, D] for a vectorized one ([R, V, D] -> ([R, V], [R, V, D]),
    column v being function v).  Returns (successes, values, points, nfev) shaped [R(, V)(, D)]."""
    import greenlet

    starts = np.asarray(starting_points, dtype=np.float64)
    vectorized = starts.ndim == 3
    lead = starts.shape[:-1]
    D = starts.shape[-1]
    flat = starts.reshape(-1, D)
    R = flat.shape[0]
    lower = np.broadcast_to(np.asarray(space.lower, dtype=np.float64), (D,))
    upper = np.broadcast_to(np.asarray(space.upper, dtype=np.float64), (D,))
    bounds = spo.Bounds(lower, upper)
    args = dict(optimizer_args or {})
    for forbidden in ("method", "jac", "bounds"):
        if forbidden in args:
            raise ValueError(f"optimizer_args must not set {forbidden!r}")
    opts = _lockstep_options(args)
    if opts is not None:
        scratch = flat.copy()   # (a vectorized function is evaluated on the full tensor, finished slots idle)

        def evaluate(xs, rows):
            if vectorized:
                scratch[rows] = xs
                vals, grads = fn.value_and_gradient(scratch.reshape(starts.shape))
                vals, grads = -_to_host(vals).reshape(R)[rows], -_to_host(grads).reshape(R, D)[rows]
            else:
                vals, grads = fn.value_and_gradient(xs)
                vals, grads = -_to_host(vals), -_to_host(grads)
            return vals, grads

        ok, fmin, xs, nfev = _lockstep_lbfgsb(evaluate, flat, lower, upper, opts)
        return ok.reshape(lead), (-fmin).reshape(lead), xs.reshape(starts.shape), nfev.reshape(lead)

    class _Run(greenlet.greenlet):
        def run(self, start):
            seen = {"x": None, "f": None, "g": None}

            def value_and_gradient(x):
                if seen["x"] is None or not np.array_equal(seen["x"], x):
                    seen["x"] = np.array(x, dtype=np.float64)
                    seen["f"], seen["g"] = self.parent.switch(seen["x"])
                return seen["f"], seen["g"]

            return spo.minimize(lambda x: value_and_gradient(x)[0], start, jac=lambda x: value_and_gradient(x)[1],
                                bounds=bounds, method="L-BFGS-B", **args)

    runs = [_Run() for _ in range(R)]
    pending = [run.switch(flat[i]) for i, run in enumerate(runs)]
    batch_x = flat.copy()
    while True:
        active = [i for i, res in enumerate(pending) if not isinstance(res, spo.OptimizeResult)]
        if not active:
            break
        for i in active:
            batch_x[i] = pending[i]
        if vectorized:  # slot (r, v) belongs to function v: evaluate the full tensor, finished slots idle
            vals, grads = fn.value_and_gradient(batch_x.reshape(starts.shape))
            vals, grads = -_to_host(vals).reshape(R), -_to_host(grads).reshape(R, D)
            vals, grads = vals[active], grads[active]
        else:
            vals, grads = fn.value_and_gradient(batch_x[active])
            vals, grads = -_to_host(vals), -_to_host(grads)
        for j, i in enumerate(active):
            if runs[i].dead:  # a crashed run is skipped, like the reference does
                continue
            pending[i] = runs[i].switch(float(vals[j]), np.array(grads[j], dtype=np.float64))
    successes = np.array([bool(r.success) for r in pending]).reshape(lead)
    values = np.array([-float(r.fun) for r in pending]).reshape(lead)
    points = np.stack([np.asarray(r.x, dtype=np.float64) for r in pending]).reshape(starts.shape)
    nfev = np.array([int(r.nfev) for r in pending]).reshape(lead)
    return successes, values, points, nfev


def generate_continuous_optimizer(num_initial_samples=NUM_SAMPLES_MIN, num_optimization_runs: int = 10,
                                  num_recovery_runs: int = 10, optimizer_args: Optional[Dict[str, Any]] = None):
    """Gradient-based optimizer for a Box and batches of size one (optimizer.py:344-560): sweep
    ``num_initial_samples`` random points (or the batches of a sampler callable), start L-BFGS-B from
    the best ``num_optimization_runs`` of them, fall back to ``num_recovery_runs`` random starts if
    every run fails, raise :class:`FailedOptimizationError` if those fail too.  The acquisition
    function must expose ``value_and_gradient`` (the engine-backed EI / PI / -LCB do)."""
    if num_optimization_runs <= 0:
        raise ValueError(f"num_optimization_runs must be positive, got {num_optimization_runs}")
    if not callable(num_initial_samples) and num_initial_samples < num_optimization_runs:
        raise ValueError(f"num_initial_samples {num_initial_samples} must be at least num_optimization_runs "
                         f"{num_optimization_runs}")
    if num_recovery_runs < 0:
        raise ValueError(f"num_recovery_runs must be zero or greater, got {num_recovery_runs}")

    def optimize_continuous(space: Box, target_func) -> np.ndarray:
        fn, V = _split(target_func)
        if V <= 0:
            raise ValueError(f"vectorization must be positive, got {V}")
        if not isinstance(space, Box):
            raise NotImplementedError("the continuous optimizer supports Box search spaces")
        if not hasattr(fn, "value_and_gradient"):
            raise TypeError("generate_continuous_optimizer needs an acquisition function exposing "
                            "value_and_gradient (there is no autodiff on this engine)")
        vectorized = isinstance(target_func, tuple)
        sampler = num_initial_samples if callable(num_initial_samples) else sample_from_space(num_initial_samples)
        initial_points = generate_initial_points(num_optimization_runs, sampler, space, fn, V)  # [k, V, D]
        if len(initial_points) < num_optimization_runs:
            raise ValueError(f"Not enough initial points generated ({len(initial_points)} for "
                             f"{num_optimization_runs} optimization runs)")
        starts = initial_points if vectorized else initial_points[:, 0, :]
        successes, values, points, _ = _perform_parallel_continuous_optimization(fn, space, starts, optimizer_args or {})
        ok = bool(np.all(np.any(successes, axis=0)))  # at least one successful run for each function
        if num_recovery_runs and not ok:
            random_points = np.asarray(space.sample(num_recovery_runs), dtype=np.float64)
            if vectorized:
                random_points = np.tile(random_points[:, None, :], [1, V, 1])  # [num_recovery_runs, V, D]
            rs, rv, rp, _ = _perform_parallel_continuous_optimization(fn, space, random_points, optimizer_args or {})
            successes, values, points = (np.concatenate([successes, rs]), np.concatenate([values, rv]),
                                         np.concatenate([points, rp]))
            ok = bool(np.all(np.any(successes, axis=0)))
        if not ok:
            raise FailedOptimizationError(f"Acquisition function optimization failed, even after "
                                          f"{num_recovery_runs + num_optimization_runs} restarts.")
        if not vectorized:
            return points[int(np.argmax(values))][None, :]
        best = np.argmax(values, axis=0)  # [V]
        return points[best, np.arange(V), :]  # [V, D]

    return optimize_continuous


class _FlattenedBatchFunction:
    """``x [..., 1, B * D] -> f(x reshaped to [..., B, D])`` (optimizer.py:921-926) for a batch function that exposes
    ``value_and_gradient([P, B, D]) -> ([P], [P, B, D])``: the flattened function's is ``[P, B * D] -> ([P], [P, B * D])``."""

    def __init__(self, f, batch_size: int):
        self._f, self._batch_size = f, batch_size

    def __call__(self, x):
        return self._f(x.reshape(tuple(x.shape[:-2]) + (self._batch_size, -1)))

    def value_and_gradient(self, points):
        pts = _to_host(points)
        vals, grads = self._f.value_and_gradient(pts.reshape(pts.shape[0], self._batch_size, -1))
        return _to_host(vals), _to_host(grads).reshape(pts.shape)


def batchify_joint(batch_size_one_optimizer, batch_size: int):
    """Optimise the B points of a batch acquisition function jointly over space**B
    (optimizer.py:897-934)."""
    if batch_size <= 0:
        raise ValueError(f"batch_size must be positive, got {batch_size}")

    def optimizer(search_space: SearchSpace, f) -> np.ndarray:
        if isinstance(f, tuple):
            raise ValueError("batchify_joint cannot be applied to a vectorized acquisition function")
        expanded = search_space ** batch_size

        def target_func_with_vectorized_inputs(x):  # [..., 1, B * D] -> [..., 1]
            return f(x.reshape(tuple(x.shape[:-2]) + (batch_size, -1)))

        if hasattr(f, "value_and_gradient") and hasattr(getattr(f, "_engine", None), "joint_vjp"):
            # a batch function with an analytic gradient (qEI): the flattened function carries it, so that
            # automatic_optimizer_selector / generate_continuous_optimizer refine with L-BFGS-B as the reference does
            target_func_with_vectorized_inputs = _FlattenedBatchFunction(f, batch_size)
        pts = batch_size_one_optimizer(expanded, target_func_with_vectorized_inputs)  # [1, B * D]
        return np.asarray(pts).reshape(batch_size, -1)

    return optimizer


def batchify_vectorize(batch_size_one_optimizer, batch_size: int):
    """Optimise V independent functions at once (optimizer.py:937-970)."""
    if batch_size <= 0:
        raise ValueError(f"batch_size must be positive, got {batch_size}")

    def optimizer(search_space: SearchSpace, f) -> np.ndarray:
        if isinstance(f, tuple):
            raise ValueError("batchify_vectorize cannot be applied to an already vectorized acquisition function")
        return batch_size_one_optimizer(search_space, (f, batch_size))

    return optimizer
