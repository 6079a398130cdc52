"""Seeding of the host-side random draws (candidate samples, eps, RFF bases, trajectory weights, prior
draws).  The reference draws through TensorFlow's global generator, which its tests pin with
``tf.random.set_seed`` (tests/util/misc.py ``random_seed``); here every draw comes from a numpy
``Generator`` made by :func:`make_rng` -- an explicit ``seed`` wins, otherwise a child of the seed
sequence installed by :func:`set_seed`, otherwise fresh OS entropy."""
from __future__ import annotations

import threading
from typing import Optional

import numpy as np

_lock = threading.Lock()
_sequence: Optional[np.random.SeedSequence] = None


def set_seed(seed: Optional[int]) -> None:
    """Make every subsequent un-seeded draw of the package reproducible (``None`` restores OS entropy)."""
    global _sequence
    with _lock:
        _sequence = None if seed is None else np.random.SeedSequence(seed)


def make_rng(seed: Optional[int] = None) -> np.random.Generator:
    if seed is not None:
        return np.random.default_rng(seed)
    with _lock:
        if _sequence is None:
            return np.random.default_rng()
        return np.random.default_rng(_sequence.spawn(1)[0])
