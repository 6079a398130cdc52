"""Multi-GPU paths on whatever GPUs the box has (one on the test box: every collective then runs at world size 1;
with more devices the same tests shard over all of them):

* the single-controller group of the C-ABI (tgp_group_*): replicated update, sharded candidates, in-process RCCL
  all-gather (and the peer-copy merge) -- winners bit-identical to the single-handle calls;
* the device-resident winner protocol over torch.distributed "nccl" (= RCCL): tgp_acq_argmax_async ->
  all_gather_into_tensor -> tgp_merge_winners_async -> one copy;
* bench.py refuses to print an N-GPU line from fewer ranks / devices, and spawns its own ranks.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(N=300, d=6, kind="rbf", noise=1e-2):
    X, Y = O.synthetic_problem(O.hartmann_6, d, N)
    return X, Y, O.default_lengthscales(d), float(np.mean(Y)), kind, noise


def _single(X, Y, ls, c, kind, noise):
    from trieste_amd.engine import GPEngine

    eng = GPEngine(X.shape[1], kind)
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    return eng


def _device_sets():
    import torch

    n = torch.cuda.device_count()
    sets = [[0]]
    if n >= 2:
        sets.append(list(range(n)))
        sets.append([1, 0])
    return sets


@pytest.mark.parametrize("merge", ["rccl", "peer"])
def test_group_winners_equal_the_single_handle(merge):
    from trieste_amd.group import GPEngineGroup

    X, Y, ls, c, kind, noise = _problem()
    eng = _single(X, Y, ls, c, kind, noise)
    eta = eng.eta()
    M = 20011
    for devices in _device_sets():
        grp = GPEngineGroup(X.shape[1], kind, devices=devices, merge=merge)
        info = grp.info()
        assert info["n_dev"] == len(devices) and info["merge"] == merge
        assert info["rccl_ranks"] == (len(devices) if merge == "rccl" else 0)
        grp.set_hyper(1.0, ls, noise, c)
        grp.set_data(X, Y)
        assert grp.eta() == eta
        # replicas are bit-identical
        L0 = grp.members[0].get_factor()
        for m in grp.members[1:]:
            for a, b in zip(L0, m.get_factor()):
                np.testing.assert_array_equal(a, b)
        # (1) one logical Philox table generated shard by shard on the devices
        grp.sample_candidates(5678, M, 0.0, 1.0)
        Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
        want = eng.acq_argmax("ei", eta, Xq)
        got = grp.acq_argmax("ei", eta)
        assert (got[0], got[1]) == (want[0], want[1])
        np.testing.assert_array_equal(got[2], want[2])
        tv, ti = grp.acq_topk("ei", eta, 33)
        wv, wi = eng.acq_topk("ei", eta, Xq, 33)
        np.testing.assert_array_equal(ti, wi)
        np.testing.assert_array_equal(tv, wv)
        # (2) a host table scattered over the members, with a tie across what would be shard boundaries
        pts = np.random.default_rng(1).uniform(size=(1001, X.shape[1]))
        pts[900] = pts[17]
        grp.set_candidates(pts)
        for acq, param in (("ei", eta), ("pi", eta), ("nlcb", 1.96)):
            want = eng.acq_argmax(acq, param, pts)
            got = grp.acq_argmax(acq, param)
            assert (got[0], got[1]) == (want[0], want[1]), acq
        # (3) trajectories: replicated weights, sharded arg-min
        rng = np.random.default_rng(11)
        F, B = 64, 3
        draws = (rng.standard_normal((F, X.shape[1])), rng.uniform(0, 2 * np.pi, F), rng.standard_normal((F, B)),
                 rng.standard_normal((X.shape[0], B)))
        wv, wi = eng.trajectory(*draws).argmin(pts)
        gv, gi = grp.trajectory(*draws).argmin()
        np.testing.assert_array_equal(gi, wi)
        np.testing.assert_array_equal(gv, wv)
        # (4) qEI sharded over the q-batches
        Xg = rng.uniform(size=(37, 5, X.shape[1]))
        eps = rng.standard_normal((5, 64))
        np.testing.assert_array_equal(grp.qei(Xg, eps, eta), eng.qei(Xg, eps, eta))
        # (5) rank-k append on every replica
        Xn, Yn = rng.uniform(size=(3, X.shape[1])), rng.standard_normal(3)
        grp.append_data(Xn, Yn)
        eng2 = _single(np.concatenate([X, Xn]), np.concatenate([Y, Yn]), ls, c, kind, noise)
        g2 = grp.acq_argmax("ei", eta)
        w2 = eng2.acq_argmax("ei", eta, pts)
        assert g2[1] == w2[1] and abs(g2[0] - w2[0]) <= 1e-9 * abs(w2[0]) + 1e-13
        grp.close()


def test_group_under_auto_precision_returns_the_float64_winner(monkeypatch):
    """TGP_PREC_AUTO on a group (also three members sharing the one GPU): every member runs the int8 sweep with the
    float64 repair on its shard, enqueue-only; the merged winner is the float64 sweep's (index; value to the summation
    order of the repair's row-group split), for several acquisition functions."""
    from trieste_amd.group import GPEngineGroup

    X, Y, ls, c, kind, noise = _problem()
    eng = _single(X, Y, ls, c, kind, noise)
    eta = eng.eta()
    M = 20011
    Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
    monkeypatch.setenv("TGP_GROUP_ALLOW_DUPLICATES", "1")
    for devices, merge in (([0], "rccl"), ([0, 0, 0], "peer")):
        grp = GPEngineGroup(X.shape[1], kind, devices=devices, merge=merge)
        grp.set_hyper(1.0, ls, noise, c)
        grp.set_data(X, Y)
        grp.set_precision("auto")
        grp.sample_candidates(5678, M, 0.0, 1.0)
        for acq, param in (("ei", eta), ("ei", float(np.median(Y))), ("pi", eta), ("nlcb", 1.96)):
            want = eng.acq_argmax(acq, param, Xq)
            for _ in range(2):          # (the second call runs after the first one's report has come back)
                got = grp.acq_argmax(acq, param)
                assert got[1] == want[1] and abs(got[0] - want[0]) <= 1e-12 * abs(want[0]) + 1e-300, (acq, got, want)
        assert all(m.get_precision()[1] in ("i8x4", "i8x5", "f64") for m in grp.members)
        grp.close()


def test_three_members_on_one_gpu_equal_the_single_handle(monkeypatch):
    """The whole multi-member path (worker threads, contiguous shards with index_base, per-member streams, peer copies
    and events, the 3-way merge kernel, host merge of the top-k, sharded trajectories and q-batches) on a single-GPU
    box: three members share device 0 (test aid TGP_GROUP_ALLOW_DUPLICATES, peer merge)."""
    from trieste_amd.group import GPEngineGroup

    monkeypatch.setenv("TGP_GROUP_ALLOW_DUPLICATES", "1")
    X, Y, ls, c, kind, noise = _problem()
    eng = _single(X, Y, ls, c, kind, noise)
    eta = eng.eta()
    grp = GPEngineGroup(X.shape[1], kind, devices=[0, 0, 0], merge="peer")
    assert grp.info()["n_dev"] == 3
    grp.set_hyper(1.0, ls, noise, c)
    grp.set_data(X, Y)
    for m in grp.members[1:]:
        for a, b in zip(grp.members[0].get_factor(), m.get_factor()):
            np.testing.assert_array_equal(a, b)
    M = 50021  # ragged three-way shards
    grp.sample_candidates(5678, M, 0.0, 1.0)
    Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
    want, got = eng.acq_argmax("ei", eta, Xq), grp.acq_argmax("ei", eta)
    assert (got[0], got[1]) == (want[0], want[1])
    np.testing.assert_array_equal(got[2], want[2])
    tv, ti = grp.acq_topk("ei", eta, 41)
    wv, wi = eng.acq_topk("ei", eta, Xq, 41)
    np.testing.assert_array_equal(ti, wi)
    np.testing.assert_array_equal(tv, wv)
    pts = np.random.default_rng(1).uniform(size=(1000, X.shape[1]))
    pts[700] = pts[17]            # a tie between the first and the last shard: the first index wins
    pts[350] = pts[17]
    grp.set_candidates(pts)
    for acq, param in (("ei", eta), ("pi", eta), ("nlcb", 1.96), ("aei", eta)):
        want, got = eng.acq_argmax(acq, param, pts), grp.acq_argmax(acq, param)
        assert (got[0], got[1]) == (want[0], want[1]), acq
    grp.set_candidates(pts[:2])   # fewer candidates than members: an empty shard never wins
    want, got = eng.acq_argmax("ei", eta, pts[:2]), grp.acq_argmax("ei", eta)
    assert (got[0], got[1]) == (want[0], want[1])
    grp.set_candidates(pts)
    rng = np.random.default_rng(11)
    F, B = 64, 5
    draws = (rng.standard_normal((F, X.shape[1])), rng.uniform(0, 2 * np.pi, F), rng.standard_normal((F, B)),
             rng.standard_normal((X.shape[0], B)))
    wv, wi = eng.trajectory(*draws).argmin(pts)
    gv, gi = grp.trajectory(*draws).argmin()
    np.testing.assert_array_equal(gi, wi)
    np.testing.assert_array_equal(gv, wv)
    Xg = rng.uniform(size=(100, 7, X.shape[1]))
    eps = rng.standard_normal((7, 32))
    np.testing.assert_array_equal(grp.qei(Xg, eps, eta), eng.qei(Xg, eps, eta))
    grp.close()


def test_group_errors_are_reported():
    from trieste_amd.group import GPEngineGroup

    grp = GPEngineGroup(2, "matern52", devices=[0])
    with pytest.raises(RuntimeError):
        grp.acq_argmax("ei", 0.0)            # no candidates yet
    grp.set_hyper(1.0, [0.5, 0.5], 1e-3, 0.0)
    grp.set_candidates(np.zeros((4, 2)))
    with pytest.raises(RuntimeError):
        grp.acq_argmax("ei", 0.0)            # no data yet (member status surfaces through the group)
    with pytest.raises(ValueError):
        GPEngineGroup(2, "matern52", devices=[0, 0])
    with pytest.raises(ValueError):
        GPEngineGroup(2, "matern52", devices=[10 ** 6])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_device_resident_winners_over_nccl_world_size_1():
    """The RCCL path of the one-process-per-GPU form, at the world size this box allows."""
    import torch
    import torch.distributed as dist

    from trieste_amd.distributed import all_gather_winners

    X, Y, ls, c, kind, noise = _problem()
    eng = _single(X, Y, ls, c, kind, noise)
    eng.use_torch_stream()
    eta = eng.eta()
    Xq = eng.sample_box(5678, 1000, 30011, 0.0, 1.0)
    want = eng.acq_argmax("ei", eta, Xq, index_base=1000)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        assert dist.get_backend() == "nccl"
        pair = eng.acq_argmax_pair("ei", eta, Xq, index_base=1000)
        assert pair.is_cuda and pair.shape == (2,)
        v, i = all_gather_winners(eng, pair)
        assert (float(v[0]), int(i[0])) == (want[0], want[1])
        rng = np.random.default_rng(5)
        F, B = 48, 4
        traj = eng.trajectory(rng.standard_normal((F, X.shape[1])), rng.uniform(0, 6.28, F),
                              rng.standard_normal((F, B)), rng.standard_normal((X.shape[0], B)))
        wv, wi = traj.argmin(Xq, index_base=1000)
        tv, ti = all_gather_winners(eng, traj.argmin_pairs(Xq, index_base=1000), minimize=True)
        np.testing.assert_array_equal(ti, wi)
        np.testing.assert_array_equal(tv, wv)
    finally:
        dist.destroy_process_group()


def test_merge_kernel_semantics():
    """tgp_merge_winners_async == distributed.merge_best: ties -> smaller global index, NaN / empty never win."""
    import torch

    from trieste_amd.distributed import merge_best
    from trieste_amd.engine import GPEngine

    eng = GPEngine(2, "rbf")
    vals = np.array([[1.0, 5.0, np.nan], [1.0, 5.0, np.nan], [0.5, np.nan, np.nan], [np.nan, 7.0, 1.0]])
    idxs = np.array([[40, 7, 3], [3, 9, 4], [1, 2, 5], [0, -1, -1]], dtype=np.int64)
    g = torch.from_numpy(np.stack([vals, idxs.view(np.float64)], axis=1)).cuda()      # [P, 2, V]
    for minimize in (False, True):
        out = eng.merge_winners(g, minimize).cpu()
        v = out[0].numpy()
        i = out[1].contiguous().view(torch.int64).numpy()
        wv, wi = merge_best(vals, idxs, minimize)
        np.testing.assert_array_equal(i[:2], wi[:2])
        np.testing.assert_array_equal(v[:2], wv[:2])
        assert np.isnan(v[2]) and i[2] == -1          # nothing valid anywhere


def _run(cmd, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_to_misreport_the_gpu_count():
    import torch

    n = torch.cuda.device_count()
    r = _run([sys.executable, "bench.py", "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)
    env_cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"]
    r = _run(env_cmd)
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)


@pytest.mark.parametrize("how", ["torchrun", "group", "selfspawn"])
def test_bench_prints_the_ranks_it_ran(how):
    import torch

    n = torch.cuda.device_count()
    common = ["--steps", "1", "--warmup", "0", "--workload", "c2", "--m-per-gpu", "20000", "--no-cpu-baseline", "--no-acquire"]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "1"] + common
        want_n, want_ranks = 1, 1
    elif how == "group":
        cmd = [sys.executable, "bench.py", "--gpus", str(n), "--mode", "group"] + common
        want_n, want_ranks = n, n
    else:
        if n < 2:
            pytest.skip("self-spawn needs >= 2 GPUs (the refusal on fewer is tested above)")
        cmd = [sys.executable, "bench.py", "--gpus", str(n)] + common
        want_n, want_ranks = n, n
    r = _run(cmd)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == want_n and out["config"]["rccl_ranks"] == want_ranks
    assert out["value"] > 0 and out["roofline"]["kernel_ms"] > 0
    per_rank = out["config"]["kernel_ms_per_rank"]  # the load balance of a multi-GPU line: one entry per rank / member
    assert len(per_rank) == want_n and all(t > 0 for t in per_rank)


def test_scale_check_script_at_the_gpus_that_exist(tmp_path):
    """tools/scale_check.sh (the one-command scaling run for an 8-GPU node): here over the GPU counts this box has, both
    multi-GPU forms, a FIXED total candidate set -- it must find ONE winner across the GPU counts and the two forms, and
    print the table with every rank's kernel time."""
    import torch

    n = torch.cuda.device_count()
    gpus = " ".join(str(g) for g in (1, 2, 4, 8) if g <= n)
    env = dict(os.environ, SCALE_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(["bash", "tools/scale_check.sh", "--gpus", gpus, "--workloads", "headline c5", "--steps", "1",
                        "--m-per-gpu", "4096"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "ok headline: one winner across" in r.stdout and "ok c5: one winner across" in r.stdout, r.stdout
    assert "kernel ms per rank" in r.stdout and "!!" not in r.stdout


def test_model_with_devices_through_the_host_layer():
    """GaussianProcessRegression(devices=[...]) on the real engine: EGO's sweeps shard over the group, the acquired point
    equals the single-device model's, updates are replicated, the observer is called once per step."""
    import torch

    import trieste_amd.models as M
    from trieste_amd import objectives as OBJ
    from trieste_amd.acquisition import EfficientGlobalOptimization, ExpectedImprovement, generate_random_search_optimizer
    from trieste_amd.bayesian_optimizer import BayesianOptimizer
    from trieste_amd.data import OBJECTIVE, Dataset
    from trieste_amd.space import Box

    rng = np.random.default_rng(0)
    x = rng.uniform(size=(40, 2))
    data = Dataset(x, OBJ.scaled_branin(x))
    box = Box([0.0, 0.0], [1.0, 1.0])
    devices = list(range(torch.cuda.device_count()))
    single = M.GaussianProcessRegression(M.build_gpr(data, box, likelihood_variance=1e-3))
    multi = M.GaussianProcessRegression(M.build_gpr(data, box, likelihood_variance=1e-3), devices=devices)
    assert multi.group.info()["rccl_ranks"] == len(devices)
    fs = ExpectedImprovement().prepare_acquisition_function(single, dataset=data)
    fm = ExpectedImprovement().prepare_acquisition_function(multi, dataset=data)
    pts = rng.uniform(size=(5001, 2))
    a, b = fs.argmax(pts), fm.argmax(pts)
    assert (a[0], a[1]) == (b[0], b[1])
    opt = generate_random_search_optimizer(20000, seed=11)
    np.testing.assert_array_equal(opt(box, fs), opt(box, fm))
    calls = []

    def observer(q):
        calls.append(len(q))
        return Dataset(q, OBJ.scaled_branin(q))

    res = BayesianOptimizer(observer, box).optimize(3, data, multi, EfficientGlobalOptimization(optimizer=opt),
                                                    fit_model=True, fit_initial_model=False, track_state=False)
    assert calls == [1, 1, 1] and all(m.N == 43 for m in multi.group.members)
    assert len(res.final_result.unwrap().datasets[OBJECTIVE]) == 43


def test_sharded_optimizer_keeps_winners_on_the_device():
    """generate_sharded_discrete_optimizer on the real engine (single process: the merge of one): the device-pair path
    returns the point the plain fused arg-max returns."""
    import trieste_amd.models as M
    from trieste_amd import objectives as OBJ
    from trieste_amd.acquisition import ExpectedImprovement, optimize_discrete
    from trieste_amd.data import Dataset
    from trieste_amd.distributed import generate_sharded_discrete_optimizer
    from trieste_amd.space import Box, DiscreteSearchSpace

    rng = np.random.default_rng(0)
    x = rng.uniform(size=(30, 2))
    data = Dataset(x, OBJ.scaled_branin(x))
    model = M.GaussianProcessRegression(M.build_gpr(data, Box([0, 0], [1, 1]), likelihood_variance=1e-3))
    fn = ExpectedImprovement().prepare_acquisition_function(model, dataset=data)
    space = DiscreteSearchSpace(rng.uniform(size=(4097, 2)))
    np.testing.assert_array_equal(generate_sharded_discrete_optimizer()(space, fn), optimize_discrete(space, fn))
