"""Property tests (hypothesis) of the host-side sweep semantics: sharding, winner merge, running top-k,
call splitting -- the rules of reference acquisition/optimizer.py:124-170, 247-341 and
acquisition/utils.py:31-80 that every sharded / chunked evaluation must preserve."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from trieste_amd.acquisition import generate_initial_points, split_acquisition_function
from trieste_amd.distributed import merge_best, shard_range
from trieste_amd.space import Box


@given(M=st.integers(0, 5000), world=st.integers(1, 9))
def test_shard_range_is_an_ordered_partition(M, world):
    spans = [shard_range(M, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == M
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert a <= b == c <= d
    assert max(b - a for a, b in spans) <= -(-M // world) if M else True


@given(data=st.data())
@settings(max_examples=60, deadline=None)
def test_sharded_argmax_equals_global_first_index_argmax(data):
    """arg-max over shards + merge == np.argmax (first index on ties), for any sharding, with ties and NaNs."""
    M = data.draw(st.integers(1, 200))
    world = data.draw(st.integers(1, 7))
    vals = np.array(data.draw(st.lists(st.sampled_from([0.0, 1.0, 2.5, -1.0, float("nan")]), min_size=M, max_size=M)))
    shard_v, shard_i = [], []
    for r in range(world):
        lo, hi = shard_range(M, r, world)
        v = vals[lo:hi]
        ok = ~np.isnan(v)
        if hi == lo or not ok.any():
            shard_v.append(float("nan")); shard_i.append(-1)
        else:
            j = int(np.argmax(np.where(ok, v, -np.inf)))
            shard_v.append(float(v[j])); shard_i.append(lo + j)
    gv, gi = merge_best(np.array(shard_v), np.array(shard_i))
    if np.all(np.isnan(vals)):
        return
    want = int(np.argmax(np.where(np.isnan(vals), -np.inf, vals)))
    assert int(gi[0]) == want and gv[0] == vals[want]


@given(seed=st.integers(0, 10_000), n_batches=st.integers(1, 5), k=st.integers(1, 12))
@settings(max_examples=30, deadline=None)
def test_running_top_k_equals_top_k_of_everything(seed, n_batches, k):
    rng = np.random.default_rng(seed)
    box = Box([0.0, 0.0], [1.0, 1.0])
    batches = [np.round(rng.uniform(size=(rng.integers(1, 30), 2)), 1) for _ in range(n_batches)]  # many ties

    def fn(x):  # [M, 1, 2] -> [M, 1]
        return (x[..., 0] - 0.3) ** 2 + x[..., 1]

    got = generate_initial_points(k, lambda space: iter(batches), box, fn)
    allp = np.concatenate(batches)
    vals = fn(allp[:, None, :])[:, 0]
    order = np.lexsort((np.arange(len(vals)), -vals))[: min(k, len(vals))]
    np.testing.assert_array_equal(got[:, 0, :], allp[order])


@given(M=st.integers(1, 300), split=st.integers(1, 700), seed=st.integers(0, 100))
@settings(max_examples=40, deadline=None)
def test_split_calls_are_equivalent_to_one_call(M, split, seed):
    x = np.random.default_rng(seed).uniform(size=(M, 1, 3))
    calls = []

    def fn(z):
        calls.append(z.shape[0])
        return np.sum(z, axis=-1)

    np.testing.assert_array_equal(split_acquisition_function(fn, split)(x), fn(x))
    assert max(calls[:-1]) <= max(1, -(-split // 3))  # chunk length = ceil(split / elements per row)
