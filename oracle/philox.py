"""Philox4x32-10 counter-based generator and the engine's uniform-candidate map.  TEST INFRASTRUCTURE ONLY.

The reference draws its candidates with ``Box.sample`` (trieste/space.py:843-867 -> ``tf.random.uniform``); TensorFlow's
own Philox stream is not reproducible outside TensorFlow, so the engine defines its own device-side draw
(``tgp_sample_box``, include/tgp.h): element (row r, column c) of a logical [M, d] sample is

    u = philox4x32_10(counter = (lo32(e), hi32(e), 0x7467705f, 0), key = (lo32(seed), hi32(seed)))   e = r * d + c
    x = lower_c + (upper_c - lower_c) * ((u[0] << 32 | u[1]) >> 11) * 2^-53

This module restates that integer path in numpy (vectorised uint64 arithmetic) so the device kernel can be checked
BIT-EXACTLY.  The block function follows Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3"
(SC'11) / the Random123 library (philox.h: multipliers 0xD2511F53, 0xCD9E8D57; Weyl key increments 0x9E3779B9,
0xBB67AE85; 10 rounds) and is pinned to Random123's published known-answer vectors (tests/golden/philox_kat.json).
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint64(0x9E3779B9)
W1 = np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
CANDIDATE_STREAM = 0x7467705F  # third counter word of the engine's candidate stream ("tgp_")


def philox4x32_10(counter, key):
    """counter: 4 arrays (or scalars) of 32-bit words, key: 2 -> 4 uint64 arrays holding 32-bit words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in counter)
    k0, k1 = (np.asarray(k, dtype=np.uint64) & MASK for k in key)
    s32 = np.uint64(32)
    for _ in range(10):
        p0 = M0 * c0  # 32 x 32 -> 64 bit products
        p1 = M1 * c2
        n0 = (p1 >> s32) ^ c1 ^ k0
        n1 = p1 & MASK
        n2 = (p0 >> s32) ^ c3 ^ k1
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def uniform53(seed: int, elem) -> np.ndarray:
    """The engine's uniform in [0, 1) with 53 random bits for element index ``elem`` (uint64 array)."""
    e = np.asarray(elem, dtype=np.uint64)
    seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    s32 = np.uint64(32)
    o0, o1, _, _ = philox4x32_10((e & MASK, e >> s32, np.uint64(CANDIDATE_STREAM), np.uint64(0)),
                                 (seed & MASK, seed >> s32))
    bits = (o0 << s32) | o1
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sample_box(seed: int, first: int, M: int, lower, upper) -> np.ndarray:
    """Rows [first, first + M) of the logical sample: [M, d] float64, bit-identical to tgp_sample_box."""
    lower = np.atleast_1d(np.asarray(lower, dtype=np.float64))
    upper = np.atleast_1d(np.asarray(upper, dtype=np.float64))
    d = lower.shape[0]
    rows = np.arange(first, first + M, dtype=np.uint64)[:, None]
    elem = rows * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]
    u = uniform53(seed, elem)
    return lower[None, :] + (upper - lower)[None, :] * u
