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


# ---- greedy-batch / entropy pieces: properties the reference's formulas imply ------------------------------
@given(data=st.data())
@settings(max_examples=60, deadline=None, derandomize=True)
def test_asynchronous_rule_state_add_then_remove_round_trips(data):
    """rule.py:426-489: removing the points just added restores the state; removal keeps the order of the rest
    and takes out one occurrence per requested removal."""
    from trieste_amd.extras import AsynchronousRuleState

    d = data.draw(st.integers(1, 3))
    row = st.lists(st.sampled_from([0.0, 1.0, 2.0]), min_size=d, max_size=d)
    pending = np.array(data.draw(st.lists(row, min_size=0, max_size=6)), dtype=float).reshape(-1, d)
    new = np.array(data.draw(st.lists(row, min_size=1, max_size=4)), dtype=float).reshape(-1, d)
    state = AsynchronousRuleState(pending if len(pending) else None)
    grown = state.add_pending_points(new)
    assert len(grown.pending_points) == len(pending) + len(new)
    back = grown.remove_points(new)
    remaining = back.pending_points if back.has_pending_points else np.zeros((0, d))
    assert len(remaining) == len(pending)
    # as multisets the original pending points survive
    assert sorted(map(tuple, remaining)) == sorted(map(tuple, pending))
    # removing something absent changes nothing
    absent = np.full((1, d), 7.0)
    same = grown.remove_points(absent)
    np.testing.assert_array_equal(same.pending_points, grown.pending_points)


@given(data=st.data())
@settings(max_examples=50, deadline=None, derandomize=True)
def test_local_penalizers_are_probabilities_that_vanish_at_pending_points_and_saturate_far_away(data):
    from oracle import gp_oracle as O  # the restated formulas (greedy_batch.py:341-354, 376-389)

    d = data.draw(st.integers(1, 4))
    P = data.draw(st.integers(1, 5))
    rng = np.random.default_rng(data.draw(st.integers(0, 10_000)))
    pending = rng.uniform(size=(P, d))
    radius, scale = rng.uniform(0.05, 0.5, P), rng.uniform(0.01, 0.3, P)
    x = rng.uniform(-1, 2, size=(40, d))
    for kind in ("soft", "hard"):
        phi = O.PENALIZERS[kind](x, pending, radius, scale)
        assert np.all((phi >= 0.0) & (phi <= 1.0))
        far = O.PENALIZERS[kind](pending[:1] + 1e3, pending, radius, scale)
        assert far[0] > 1.0 - 1e-9
        at = O.PENALIZERS[kind](pending, pending, radius, scale)
        assert np.all(at <= 0.5 + 1e-12) if kind == "soft" else np.all(at == 0.0)
        # moving away from a lone pending point never lowers the penalization factor
        t = np.sort(rng.uniform(0, 3, 12))
        ray = pending[:1] + t[:, None] * np.ones((1, d)) / np.sqrt(d)
        mono = O.PENALIZERS[kind](ray, pending[:1], radius[:1], scale[:1])
        assert np.all(np.diff(mono) >= -1e-15)


@given(data=st.data())
@settings(max_examples=50, deadline=None, derandomize=True)
def test_entropy_tails_are_nonnegative_and_the_log_cdf_is_continuous_at_its_branch_points(data):
    from oracle import gp_oracle as O  # entropy.py:195-214, 479-500; tfp log_ndtr branches

    rng = np.random.default_rng(data.draw(st.integers(0, 10_000)))
    M, S = 30, data.draw(st.integers(1, 6))
    mean, var = rng.normal(size=M), rng.uniform(1e-6, 2.0, M)
    samples = mean.min() - rng.uniform(0.0, 1.5, S)  # minimum-value samples lie below the means
    noise = data.draw(st.sampled_from([1e-6, 1e-2, 1.0]))
    mes = O.min_value_entropy_search(mean, var, samples)
    gq = O.gibbon_quality_term(mean, var, samples, noise)
    assert np.all(mes >= -1e-12) and np.all(gq >= -1e-12)
    assert np.all(gq <= mes + 1e-9)  # rho^2 <= 1: the noisy-observation bound is the weaker one (Moss et al. 2021)
    for x0 in (-20.0, 8.0):
        lo, hi = O.log_normal_cdf(np.array([x0 - 1e-9, x0 + 1e-9]))
        assert abs(lo - hi) <= 1e-7 * max(1.0, abs(lo))
    xs = np.linspace(-40, 12, 500)
    assert np.all(np.diff(O.log_normal_cdf(xs)) > 0)  # strictly increasing through all three branches
