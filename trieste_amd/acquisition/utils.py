"""Chunking helpers (reference trieste/acquisition/utils.py:31-123)."""
from __future__ import annotations

import functools
import math

import numpy as np


def split_acquisition_function(fn, split_size: int):
    """Call ``fn`` on slices of the leading axis holding at most ``split_size`` ELEMENTS per call
    and concatenate the results (utils.py:31-80).  The fused engine streams candidate tiles and
    never materialises [N, M], so this is only needed for API parity / foreign functions."""
    if split_size <= 0:
        raise ValueError(f"split_size must be positive, got {split_size}")

    @functools.wraps(fn)
    def wrapper(x):
        length = x.shape[0]
        if length == 0:
            return fn(x)
        elements_per_block = int(np.prod(x.shape)) / length
        blocks_per_batch = int(math.ceil(split_size / elements_per_block))
        if length <= blocks_per_batch:
            return fn(x)
        outs = [fn(x[s:s + blocks_per_batch]) for s in range(0, length, blocks_per_batch)]
        if type(outs[0]).__module__.startswith("torch"):
            import torch

            return torch.cat(outs, dim=0)
        return np.concatenate(outs, axis=0)

    # the engine-backed function objects stream candidate tiles themselves: their fused sweeps, their analytic
    # gradient (the L-BFGS-B refinement of automatic_optimizer_selector) and the engine handle (on-device
    # candidate generation in generate_random_search_optimizer) pass through the wrapper unchanged
    for attr in ("argmax", "top_k", "value_and_gradient", "_engine", "_group"):
        if hasattr(fn, attr):
            setattr(wrapper, attr, getattr(fn, attr))
    return wrapper


def split_acquisition_function_calls(optimizer, split_size: int):
    """Wrap an optimizer so acquisition evaluations are chunked (utils.py:83-109)."""
    if split_size <= 0:
        raise ValueError(f"split_size must be positive, got {split_size}")

    def split_optimizer(search_space, f):
        af, n = f if isinstance(f, tuple) else (f, 1)
        taf = split_acquisition_function(af, split_size)
        return optimizer(search_space, (taf, n) if isinstance(f, tuple) else taf)

    return split_optimizer


def select_nth_output(x, output_dim: int = 0):
    """[..., B, L] -> [..., B] (utils.py:112-123)."""
    return x[..., output_dim]
