"""Multi-process (gloo, world_size 2, CPU) tests of the sharded sweep contract: contiguous shards with
index_base, one all-gather of (value, index), merge with (max value, min index)."""
import os
import socket

import numpy as np
import pytest

from trieste_amd.distributed import all_gather_best, merge_best, shard_range


def test_shard_range_partitions_and_keeps_order():
    for M in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(M, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_merge_best_tie_break_nan_and_empty():
    vals = np.array([[1.0, 5.0], [1.0, 5.0], [0.5, np.nan]])
    idxs = np.array([[40, 7], [3, 9], [1, 2]])
    v, i = merge_best(vals, idxs)
    np.testing.assert_array_equal(i, [3, 7])  # ties -> smaller global index; NaN never wins
    np.testing.assert_array_equal(v, [1.0, 5.0])
    v, i = merge_best(vals[:, :1], idxs[:, :1], minimize=True)
    assert (v[0], i[0]) == (0.5, 1)
    v, i = merge_best(np.array([-np.inf, 2.0]), np.array([-1, 11]))  # rank 0 had an empty shard
    assert (v[0], i[0]) == (2.0, 11)
    # NaN values with perfectly good indices: nothing valid -> (NaN, -1), never a NaN-valued "winner"
    v, i = merge_best(np.array([np.nan, np.nan]), np.array([3, 7]))
    assert np.isnan(v[0]) and i[0] == -1
    # a legitimate -inf value does not tie with a NaN entry of smaller index
    v, i = merge_best(np.array([np.nan, -np.inf, -np.inf]), np.array([1, 9, 5]))
    assert (v[0], i[0]) == (-np.inf, 5)
    v, i = merge_best(np.array([np.nan, np.inf]), np.array([1, 9]), minimize=True)
    assert (v[0], i[0]) == (np.inf, 9)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # one logical candidate table, sharded; every rank finds its local winner with the ORACLE
        # (the device sweep is covered by the -m gpu tests), then the product's merge runs over gloo
        from oracle import gp_oracle as O

        rng = np.random.default_rng(0)
        X, Y = O.synthetic_problem(O.branin, 2, 30)
        st = O.gpr_update("matern52", 1.0, O.default_lengthscales(2), 1e-3, 0.0, X, Y)
        eta = O.eta_min_mean(st)
        Xq = rng.uniform(size=(1001, 2))
        Xq[900] = Xq[17]  # a tie across shards: the first index must win
        lo, hi = shard_range(len(Xq), rank, world)
        vals = O.ei_values(st, Xq[lo:hi], eta)
        li = int(np.argmax(vals))
        gv, gi = all_gather_best(vals[li], lo + li)
        full = O.ei_values(st, Xq, eta)
        q.put((rank, float(gv[0]), int(gi[0]), int(np.argmax(full)), float(np.max(full))))
        # vectorised + minimise (the Thompson arg-min of B trajectories)
        tv = np.array([3.0 - rank, 1.0, 2.0 + rank])
        ti = np.array([10 * rank + 1, 5 - rank, 7 + rank])
        mv, mi = all_gather_best(tv, ti, minimize=True)
        q.put((rank, mv.tolist(), mi.tolist()))
    finally:
        dist.destroy_process_group()


def test_all_gather_best_world_size_2_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    firsts = [o for o in out if len(o) == 5]
    seconds = [o for o in out if len(o) == 3]
    assert len(firsts) == 2 and len(seconds) == 2
    for _, gv, gi, want_i, want_v in firsts:
        assert gi == want_i and gv == want_v
    for _, mv, mi in seconds:
        assert mv == [2.0, 1.0, 2.0] and mi == [11, 4, 7]


def test_single_process_is_identity():
    v, i = all_gather_best(1.5, 42)
    assert (v[0], i[0]) == (1.5, 42)


def _ego_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the product's multi-GPU EGO path on the host layer: replicated model (engine replaced by the
        # oracle-backed fake at its boundary), candidate-sharded fused arg-max, (value, index) all-gather
        import trieste_amd.models as M
        from tests.fakes import FakeEngine
        from trieste_amd import objectives as OBJ
        from trieste_amd.acquisition import EfficientGlobalOptimization, optimize_discrete
        from trieste_amd.data import Dataset
        from trieste_amd.distributed import generate_sharded_discrete_optimizer
        from trieste_amd.space import Box, DiscreteSearchSpace

        M.GPEngine = FakeEngine
        rng = np.random.default_rng(3)
        x = rng.uniform(size=(25, 2))
        data = Dataset(x, OBJ.scaled_branin(x))
        model = M.GaussianProcessRegression(M.build_gpr(data, Box([0, 0], [1, 1]), likelihood_variance=1e-3))
        cands = rng.uniform(size=(777, 2))
        cands[600] = cands[5]  # a tie across the two shards: the first index must win
        space = DiscreteSearchSpace(cands)
        rule = EfficientGlobalOptimization(optimizer=generate_sharded_discrete_optimizer())
        got = rule.acquire_single(space, model, dataset=data)
        want = EfficientGlobalOptimization(optimizer=optimize_discrete).acquire_single(space, model, dataset=data)
        q.put((rank, got.tolist(), want.tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_discrete_optimizer_world_size_2_gloo_matches_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ego_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(o[0] for o in out) == [0, 1]
    for _, got, want in out:
        assert got == want  # every rank returns the single-process winner, bit for bit
    assert out[0][1] == out[1][1]


def _greedy_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # greedy batches over a sharded candidate table: the penalized / fantasized / GIBBON acquisition functions
        # expose the same fused arg-max(points, index_base), so every batch element is one sharded sweep + all-gather
        import trieste_amd
        import trieste_amd.models as M
        from tests.fakes import FakeEngine
        from trieste_amd import objectives as OBJ
        from trieste_amd.extras import (GIBBON, EfficientGlobalOptimization, Fantasizer, GumbelSampler,
                                             LocalPenalization, optimize_discrete)
        from trieste_amd.data import Dataset
        from trieste_amd.distributed import generate_sharded_discrete_optimizer
        from trieste_amd.space import Box, DiscreteSearchSpace

        M.GPEngine = FakeEngine
        box = Box([0, 0], [1, 1])
        rng = np.random.default_rng(3)
        x = rng.uniform(size=(25, 2))
        data = Dataset(x, OBJ.scaled_branin(x))
        cands = rng.uniform(size=(501, 2))
        space = DiscreteSearchSpace(cands)
        out = {}
        for name, make in (("lp", lambda: LocalPenalization(box, num_samples=50)), ("fantasizer", lambda: Fantasizer()),
                           ("gibbon", lambda: GIBBON(box, grid_size=40, min_value_sampler=GumbelSampler(True)))):
            res = []
            for optimizer in (generate_sharded_discrete_optimizer(), optimize_discrete):
                trieste_amd.set_seed(11)  # same random draws (Lipschitz samples, Gumbel uniforms) on every rank / path
                model = M.GaussianProcessRegression(M.build_gpr(data, box, likelihood_variance=1e-3))
                rule = EfficientGlobalOptimization(make(), optimizer=optimizer, num_query_points=3)
                res.append(rule.acquire_single(space, model, dataset=data).tolist())
            out[name] = res
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_greedy_batches_over_sharded_candidates_world_size_2_gloo_match_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_greedy_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, res in out:
        for name, (sharded, single) in res.items():
            assert sharded == single, name  # every rank returns the single-process batch, bit for bit
            assert len(sharded) == 3
    assert out[0][1] == out[1][1]


def _winners_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the device-resident protocol of bench.py's step -- pair on the "device", ONE all-gather of the pairs, the
        # engine's merge, one copy to the host -- with the oracle-backed stand-in at the engine boundary
        from oracle import gp_oracle as O
        from tests.fakes import FakeEngine
        from trieste_amd.distributed import all_gather_winners

        X, Y = O.synthetic_problem(O.branin, 2, 30)
        eng = FakeEngine(2, "matern52")
        eng.set_hyper(1.0, O.default_lengthscales(2), 1e-3, 0.0)
        eng.set_data(X, Y)
        eta = eng.eta()
        M = 1001
        lo, hi = shard_range(M, rank, world)
        Xq = eng.sample_box(5678, lo, hi - lo, 0.0, 1.0)          # shard of ONE logical Philox sample
        pair = eng.acq_argmax_pair("ei", eta, Xq, index_base=lo)
        gv, gi = all_gather_winners(eng, pair)
        full = eng.sample_box(5678, 0, M, 0.0, 1.0)
        v, i, _ = eng.acq_argmax("ei", eta, full)
        q.put(("ei", rank, float(gv[0]), int(gi[0]), float(v), int(i)))
        rng = np.random.default_rng(3)
        F, B = 16, 3
        traj = eng.trajectory(rng.standard_normal((F, 2)), rng.uniform(0, 6.28, F), rng.standard_normal((F, B)),
                              rng.standard_normal((30, B)))
        tv, ti = all_gather_winners(eng, traj.argmin_pairs(Xq, index_base=lo), minimize=True)
        wv, wi = traj.argmin(full)
        q.put(("ts", rank, tv.tolist(), ti.tolist(), np.asarray(wv).tolist(), np.asarray(wi).tolist()))
    finally:
        dist.destroy_process_group()


def test_all_gather_winners_world_size_2_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_winners_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eis = [o for o in out if o[0] == "ei"]
    tss = [o for o in out if o[0] == "ts"]
    assert len(eis) == 2 and len(tss) == 2
    for _, _, gv, gi, v, i in eis:
        assert (gv, gi) == (v, i)      # every rank holds the unsharded winner, bit for bit
    for _, _, tv, ti, wv, wi in tss:
        assert tv == wv and ti == wi


def _ragged_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gp_oracle as O
        from tests.fakes import FakeEngine, _pairs
        from trieste_amd.distributed import all_gather_winners

        X, Y = O.synthetic_problem(O.branin, 2, 30)
        eng = FakeEngine(2, "matern52")
        eng.set_hyper(1.0, O.default_lengthscales(2), 1e-3, 0.0)
        eng.set_data(X, Y)
        eta = eng.eta()
        for M in (1003, 9, 3):  # M % 4 != 0; ceil(9 / 4) = 3 -> rank 3 is EMPTY; 3 points on 4 ranks -> ranks 3 empty, rows of 1
            lo, hi = shard_range(M, rank, world)
            if hi > lo:
                Xq = eng.sample_box(5678, lo, hi - lo, 0.0, 1.0)
                pair = eng.acq_argmax_pair("ei", eta, Xq, index_base=lo)
            else:  # what the sharded optimizer contributes for an empty shard: (NaN, -1), which never wins
                pair = _pairs([float("nan")], [-1])
            gv, gi = all_gather_winners(eng, pair)
            full = eng.sample_box(5678, 0, M, 0.0, 1.0)
            v, i, _ = eng.acq_argmax("ei", eta, full)
            q.put(("ragged", rank, M, (lo, hi), float(gv[0]), int(gi[0]), float(v), int(i)))
        try:  # every shard empty / invalid: an error, not a silently wrong point
            all_gather_winners(eng, _pairs([float("nan")], [-1]))
            q.put(("allempty", rank, "no error"))
        except ValueError as e:
            q.put(("allempty", rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_all_gather_winners_world_size_4_gloo_ragged_and_empty_shards():
    """SURVEY 8e at world size 4: a table that does not divide (1003), one with an EMPTY last shard (9 rows on 4
    ranks: ceil = 3 -> rank 3 owns nothing) and one with fewer rows than ranks -- every rank ends with the unsharded
    winner, bit for bit; no valid winner anywhere raises on every rank."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(16)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ragged = [o for o in out if o[0] == "ragged"]
    assert len(ragged) == 12
    for _, rank, M, (lo, hi), gv, gi, v, i in ragged:
        # (the numpy stand-in's BLAS rounds a 1-row shard differently from the 3-row table: last-ulp slack on the value;
        # the engine's values do not depend on the launch shape -- tests/test_gpu_multi.py compares them bit for bit)
        assert gi == i and abs(gv - v) <= 1e-12 * abs(v), (rank, M)
    assert any(hi == lo for _, rank, M, (lo, hi), *_ in ragged if M == 9)   # the empty shard was exercised
    empties = [o for o in out if o[0] == "allempty"]
    assert len(empties) == 4 and all("no valid winner" in o[2] for o in empties)


def test_all_gather_winners_single_process_is_the_merge_of_one():
    from tests.fakes import _pairs
    from trieste_amd.distributed import all_gather_winners

    class _E:
        def merge_winners(self, gathered, minimize=False):
            from tests.fakes import FakeEngine

            return FakeEngine.merge_winners(self, gathered, minimize)

    v, i = all_gather_winners(_E(), _pairs([1.5, -2.0], [42, 7]))
    assert v.tolist() == [1.5, -2.0] and i.tolist() == [42, 7]
