"""The integer path of candidate generation: Philox4x32-10 pinned to Random123's known-answer
vectors on the CPU, and the device kernel (tgp_sample_box) BIT-EXACT against the numpy restatement on the GPU."""
import json
import os

import numpy as np
import pytest

from oracle import philox as P

KAT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "philox_kat.json")


def test_philox4x32_10_known_answers():
    for v in json.load(open(KAT))["philox4x32_10"]:
        ctr = [int(x, 16) for x in v["counter"]]
        key = [int(x, 16) for x in v["key"]]
        out = [int(x) for x in P.philox4x32_10(ctr, key)]
        assert out == [int(x, 16) for x in v["out"]], v


def test_philox_is_vectorised_consistently():
    ctr = [np.array([0, 0xFFFFFFFF, 0x243F6A88], dtype=np.uint64), np.array([0, 0xFFFFFFFF, 0x85A308D3], dtype=np.uint64),
           np.array([0, 0xFFFFFFFF, 0x13198A2E], dtype=np.uint64), np.array([0, 0xFFFFFFFF, 0x03707344], dtype=np.uint64)]
    key = [np.array([0, 0xFFFFFFFF, 0xA4093822], dtype=np.uint64), np.array([0, 0xFFFFFFFF, 0x299F31D0], dtype=np.uint64)]
    out = np.stack(P.philox4x32_10(ctr, key), axis=1)
    want = [[int(x, 16) for x in v["out"]] for v in json.load(open(KAT))["philox4x32_10"]]
    np.testing.assert_array_equal(out, np.array(want, dtype=np.uint64))


def test_sample_box_map_properties():
    lo, up = np.array([0.0, -1.0, 2.0]), np.array([1.0, 1.0, 5.0])
    a = P.sample_box(42, 0, 5000, lo, up)
    assert a.shape == (5000, 3) and np.all(a >= lo) and np.all(a < up)
    np.testing.assert_array_equal(P.sample_box(42, 3000, 2000, lo, up), a[3000:])   # shard-consistent
    assert abs(a[:, 0].mean() - 0.5) < 0.02
    # 53-bit lattice: every uniform is k * 2^-53
    u = P.uniform53(7, np.arange(100, dtype=np.uint64))
    assert np.all(u * 2.0 ** 53 == np.floor(u * 2.0 ** 53)) and np.all((u >= 0) & (u < 1))


@pytest.mark.gpu
@pytest.mark.parametrize("d,seed,first,M", [(3, 42, 0, 10000), (8, 5678, 1_000_003, 4099), (16, 2 ** 40 + 5, 2 ** 33, 1000),
                                            (1, 0, 0, 1)])
def test_device_candidates_are_bit_exact(d, seed, first, M):
    from trieste_amd.engine import GPEngine

    eng = GPEngine(d, "rbf")
    rng = np.random.default_rng(d)
    lo = rng.uniform(-3, 0, size=d)
    up = lo + rng.uniform(0.5, 4, size=d)
    got = eng.sample_box(seed, first, M, lo, up).cpu().numpy()
    np.testing.assert_array_equal(got, P.sample_box(seed, first, M, lo, up))
    got01 = eng.sample_box(seed, first, M, 0.0, 1.0).cpu().numpy()
    np.testing.assert_array_equal(got01, P.sample_box(seed, first, M, np.zeros(d), np.ones(d)))
