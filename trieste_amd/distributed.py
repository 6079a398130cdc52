"""Candidate sharding and the cross-rank arg-max / arg-min merge.

The reference has no multi-device path.  The sweep shards naturally over
candidates: rank r evaluates the contiguous block [lo_r, hi_r) with ``index_base = lo_r`` so that
global indices -- and therefore tf.math.argmax's first-index tie-break -- are preserved, and the
only exchange is one all-gather of (value, index) pairs over RCCL/xGMI (16 B per rank and per
vectorised function), merged with the lexicographic rule (max value, min index).
Model state is replicated: every rank runs the same deterministic `update`.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(M: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ceil(M / world) candidates for `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    per = -(-M // world)
    lo = min(rank * per, M)
    return lo, min(lo + per, M)


def merge_best(vals: np.ndarray, idxs: np.ndarray, minimize: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-rank winners.  vals/idxs: [world, V].  Larger value wins (smaller if
    ``minimize``); ties go to the smaller global index; NaN / empty shards (idx < 0) never win.
    No valid entry in a column: (NaN, -1), like the device kernel (csrc merge_winners_kernel)."""
    vals = np.asarray(vals, dtype=np.float64)
    idxs = np.asarray(idxs, dtype=np.int64)
    if vals.ndim == 1:
        vals, idxs = vals[:, None], idxs[:, None]
    key = -vals if minimize else vals
    valid = ~np.isnan(vals) & (idxs >= 0) & (idxs != np.iinfo(np.int64).max)
    key = np.where(valid, key, -np.inf)
    best = np.max(key, axis=0)
    # only VALID entries may tie with the best key: a NaN value with a good index (key -inf) must not tie with a
    # legitimate -inf value, nor win a column in which nothing is valid
    cand = np.where(valid & (key == best[None, :]), idxs, np.iinfo(np.int64).max)
    win_idx = np.min(cand, axis=0)
    win_rank = np.argmin(cand, axis=0)
    win_val = vals[win_rank, np.arange(vals.shape[1])]
    none = win_idx == np.iinfo(np.int64).max
    return np.where(none, np.nan, win_val), np.where(none, -1, win_idx)


def all_gather_best(val, idx, minimize: bool = False, group=None, device=None, force: bool = False):
    """All-gather (value, index)[V] from every rank and merge.  With torch.distributed not
    initialised (single process) this is the identity.  Uses the process group's backend:
    "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
    import torch
    import torch.distributed as dist

    v = np.atleast_1d(np.asarray(val, dtype=np.float64))
    i = np.atleast_1d(np.asarray(idx, dtype=np.int64))
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return v.copy(), i.copy()
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device(device if device is not None else ("cuda" if backend == "nccl" else "cpu"))
    # one message: the int64 index travels bit-cast inside the float64 payload
    payload = torch.empty(2 * v.shape[0], dtype=torch.float64)
    payload[: v.shape[0]] = torch.from_numpy(v)
    payload[v.shape[0]:] = torch.from_numpy(i).view(torch.float64)
    payload = payload.to(dev)
    out = torch.empty(world * payload.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, payload, group=group)
    out = out.cpu().view(world, 2, v.shape[0])
    vals = out[:, 0, :].numpy()
    idxs = out[:, 1, :].contiguous().view(torch.int64).numpy()
    return merge_best(vals, idxs, minimize)


def all_gather_winners(engine, pairs, minimize: bool = False, group=None):
    """The device-resident form of :func:`all_gather_best`: ``pairs`` is what ``engine.acq_argmax_pair`` /
    ``trajectory.argmin_pairs`` left on the device ([2] or [2, V]: values, then global indices as int64 bit
    patterns).  One all-gather of the pairs (RCCL on GPUs), the merge kernel of the engine
    (``tgp_merge_winners_async``: max value -- min if ``minimize`` --, min global index), and ONE device-to-host
    copy: a single host synchronisation per sharded step instead of three.  Sweep, collective and merge are ordered
    by the stream when the engine queues on torch's current stream (``engine.use_torch_stream()``, or the default
    stream on both sides); an engine on a private stream is ordered here with host synchronisations instead (correct,
    two more host waits).  Returns (values [V], global indices [V]) as numpy arrays; identity for a single process.
    Raises ``ValueError`` when no shard produced a valid winner (every shard empty, or every value NaN)."""
    import torch
    import torch.distributed as dist

    t = pairs if _is_tensor(pairs) else torch.as_tensor(np.asarray(pairs, dtype=np.float64))
    V = int(t.numel() // 2)
    t = t.reshape(2, V)
    active = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if active else 1
    unordered = active and t.is_cuda and getattr(engine, "on_private_stream", False)
    if active:
        if unordered:
            engine.synchronize()  # the pair is written on the engine's own stream: the collective must not read it early
        out = torch.empty((world, 2, V), dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=group)
        if unordered:
            torch.cuda.current_stream(t.device).synchronize()  # ... nor the merge kernel the gathered pairs
    else:
        out = t.reshape(1, 2, V)
    merged = engine.merge_winners(out, minimize)
    host = merged.cpu() if _is_tensor(merged) else torch.as_tensor(np.asarray(merged))
    host = host.reshape(2, V)
    vals, idxs = host[0].numpy().copy(), host[1].contiguous().view(torch.int64).numpy().copy()
    if np.any(idxs < 0):
        raise ValueError("the sharded sweep produced no valid winner (every shard empty, or every value NaN)")
    return vals, idxs


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch")


def generate_sharded_discrete_optimizer(group=None, device=None):
    """An ``AcquisitionOptimizer`` (reference acquisition/optimizer.py:73-87) for one process per GPU: every
    rank holds the same model and the same ``DiscreteSearchSpace``; rank r sweeps the contiguous shard
    ``shard_range(M, r, world)`` of its points with the function's fused device arg-max (``index_base`` =
    shard offset, so global indices and the first-index tie-break survive), the (value, index) winners are
    all-gathered and merged, and every rank returns the same point [1, D].  Single process: identical to
    :func:`~trieste_amd.acquisition.optimizer.optimize_discrete`."""
    import torch.distributed as dist

    def optimizer(space, target_func):
        if isinstance(target_func, tuple):
            raise ValueError("the sharded optimizer takes batch-size-one acquisition functions")
        if not hasattr(target_func, "argmax"):
            raise TypeError("the sharded optimizer needs an engine-backed acquisition function (fused arg-max)")
        points = np.asarray(space.points, dtype=np.float64)
        active = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if active else 0
        world = dist.get_world_size(group) if active else 1
        lo, hi = shard_range(points.shape[0], rank, world)
        eng = getattr(target_func, "_engine", None)
        pair = None
        if hasattr(target_func, "argmax_pair"):  # (the choice of path must not depend on the rank: only on the function)
            try:  # device-resident winners: sweep -> all-gather of the pairs -> merge kernel -> one copy to the host
                if hi > lo:
                    pair = target_func.argmax_pair(points[lo:hi], index_base=lo)
                else:  # more ranks than points: an empty shard contributes (NaN, -1), which never wins
                    pair = target_func.argmax_pair(points[:1], index_base=0)
                    pair[0] = float("nan")
                    ints = pair.view(np.int64) if isinstance(pair, np.ndarray) else pair.view(__import__("torch").int64)
                    ints[1] = -1
            except TypeError:  # a function that installs engine state per call: host-scalar form below
                pair = None
        if pair is not None:
            _, gi = all_gather_winners(eng, pair, group=group)
            return points[int(gi[0])][None, :]
        if hi > lo:
            val, idx, _ = target_func.argmax(points[lo:hi], index_base=lo)
        else:  # more ranks than points: an empty shard never wins
            val, idx = float("nan"), -1
        _, gi = all_gather_best(val, idx, group=group, device=device)
        if int(gi[0]) < 0:
            raise ValueError("the sharded sweep produced no valid winner (every shard empty, or every value NaN)")
        return points[int(gi[0])][None, :]

    return optimizer
