"""`update` as one persistent launch (csrc/tgp_kernels_dag.hip; tgp_set_data for 512 <= Npad <= 16128 -- 4096 before round 6 --, from
Npad = 256 on with tgp_set_variant bit 5, which these tests set) on the GPU:

* L, W = L^-1 and alpha against numpy's Cholesky of the oracle's K + s I at the sizes where the task list changes
  shape (1 ... 5 block rows with and without padding, a mid size, the headline N = 4096), both noise levels;
* against the recursion of dependent launches it replaces (tgp_set_variant bit 4): the same factor up to rounding;
* BIT-IDENTICAL run to run, across handles and after a different factorisation has used the same buffers -- the
  result must not depend on which workgroup ran which task when (the replica contract of SURVEY 8e);
* a matrix that is not positive definite is reported, not hung on.
Reference: trieste/models/gpflow/models.py:171-186 -> interface.py:108-112 (K + s I, Cholesky, no jitter)."""
import numpy as np
import pytest

from oracle import gp_oracle as O
from tests.util import assert_close

pytestmark = pytest.mark.gpu

NO_DAG, DAG_SMALL, DAG_WHOLE_TILES, DAG_ONE_CHAIN = 16, 32, 256, 512


def _problem(N, d=4, kind="matern52", noise=1e-2, seed_obj=O.ackley):
    X, Y = O.synthetic_problem(seed_obj, d, N)
    ls = O.default_lengthscales(d)
    c = float(np.mean(Y))
    return X, Y, ls, c, kind, noise


def _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL):
    from trieste_amd.engine import GPEngine

    eng = GPEngine(X.shape[1], kind)
    eng.set_variant(variant)
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    return eng


@pytest.mark.parametrize("N", [300, 512, 513, 640, 1000, 1536, 2049])
@pytest.mark.parametrize("noise", [1e-2, 1e-5])
def test_dag_update_matches_numpy_and_the_recursion(N, noise):
    X, Y, ls, c, kind, _ = _problem(N, noise=noise)
    st = O.gpr_update(kind, 1.0, ls, noise, c, X, Y)
    eng = _engine(X, Y, ls, c, kind, noise)
    L, W, alpha = eng.get_factor()
    old = _engine(X, Y, ls, c, kind, noise, variant=NO_DAG)
    L0, W0, alpha0 = old.get_factor()
    scale = np.abs(st.L).max()
    cond = 1.0 + N / noise
    tol = 64 * np.finfo(float).eps * cond  # backward-stable factorisations differ by ~ eps cond(K)
    assert_close(L, st.L, rtol=1e-9, atol=tol * scale, what=f"L vs numpy (N={N})")
    assert_close(L, L0, rtol=1e-9, atol=tol * scale, what="L vs the recursion")
    Wref = np.linalg.solve(st.L, np.eye(N))
    assert_close(W, Wref, rtol=1e-7, atol=tol * np.abs(Wref).max() * 64, what="W vs numpy")
    assert_close(W, W0, rtol=1e-7, atol=tol * np.abs(Wref).max() * 64, what="W vs the recursion")
    assert np.abs(np.tril(W) @ np.tril(L) - np.eye(N)).max() < 1e-7
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L)) and np.array_equal(np.triu(W, 1), np.zeros_like(W))
    assert_close(alpha, np.linalg.solve(st.L.T, np.linalg.solve(st.L, Y - c)), rtol=1e-6,
                 atol=1e-6 * np.abs(alpha0).max(), what="alpha")
    # the posterior through the new factor
    Xq = np.random.default_rng(3).uniform(size=(257, X.shape[1]))
    Xq[:7] = X[:7]
    m, v = eng.predict(Xq)
    m0, v0 = old.predict(Xq)
    om, ov = O.predict(st, Xq)
    from tests.util import cancellation_floor

    floor = cancellation_floor(N, 1.0, noise)
    assert_close(v, ov, atol=floor, what="var through the DAG factor")
    assert_close(m, om, atol=floor, what="mean through the DAG factor")
    assert_close(v, v0, atol=floor, what="var: DAG vs recursion")


def test_dag_update_is_bit_identical_run_to_run_and_across_handles():
    X, Y, ls, c, kind, noise = _problem(4096, d=8)
    a = _engine(X, Y, ls, c, kind, noise)
    La, Wa, aa = a.get_factor()
    for _ in range(3):  # the same handle, the same buffers
        a.set_data(X, Y)
        L, W, al = a.get_factor()
        assert np.array_equal(L, La) and np.array_equal(W, Wa) and np.array_equal(al, aa)
    b = _engine(X, Y, ls, c, kind, noise)  # another handle (other buffers, other task-list upload)
    Lb, Wb, ab = b.get_factor()
    assert np.array_equal(Lb, La) and np.array_equal(Wb, Wa) and np.array_equal(ab, aa)
    # a different factorisation in between leaves nothing behind
    X2, Y2, *_ = _problem(3000, d=8, seed_obj=O.hartmann_6 if False else O.ackley)
    b.set_data(X2[:3000] * 0.9, Y2[:3000])
    b.set_data(X, Y)
    Lb, Wb, ab = b.get_factor()
    assert np.array_equal(Lb, La) and np.array_equal(Wb, Wa) and np.array_equal(ab, aa)
    # sanity at this size against numpy
    st = O.gpr_update(kind, 1.0, ls, noise, c, X, Y)
    assert_close(La, st.L, rtol=1e-9, atol=1e-9 * np.abs(st.L).max(), what="L at N = 4096")
    assert np.abs(np.tril(Wa) @ np.tril(La) - np.eye(4096)).max() < 1e-8


@pytest.mark.parametrize("N", [384, 640, 1100, 4096])
def test_split_plan_gives_the_bits_of_the_whole_tile_plan(N):
    """Round 6: at the chain-bound sizes (3 <= block rows < 48) the plan splits its two critical single products -- T(i,i-2)
    and the last burst of tile (i,i-1) -- into half-tile tasks (32 x 32 wave tiles, a fourth dependency slot, a second flag for
    the chain).  Every element is the same sum in the same order: L, W and alpha equal the whole-tile plan's (variant bit 8)
    bit for bit, run to run -- and the recursion's factor up to rounding, as before."""
    X, Y, ls, c, kind, noise = _problem(N, d=4 if N < 4096 else 8)
    split = _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL | DAG_ONE_CHAIN)
    whole = _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL | DAG_WHOLE_TILES)
    Ls, Ws, als = split.get_factor()
    Lw, Ww, alw = whole.get_factor()
    assert np.array_equal(Ls, Lw) and np.array_equal(Ws, Ww) and np.array_equal(als, alw)
    split.set_data(X, Y)
    L2, W2, al2 = split.get_factor()
    assert np.array_equal(L2, Ls) and np.array_equal(W2, Ws) and np.array_equal(al2, als)
    assert np.abs(np.tril(Ws) @ np.tril(Ls) - np.eye(N)).max() < 1e-7


def test_dag_update_at_n8200_ragged_padding():
    """The persistent kernel where it is throughput-bound (66 block rows, 36 000 tasks), with N not a multiple of the
    tile: L against numpy's factor of the oracle's K + s I, W L = I."""
    N = 8200
    X, Y, ls, c, kind, noise = _problem(N, d=8)
    eng = _engine(X, Y, ls, c, kind, noise, variant=0)   # the default policy picks the persistent form here
    L, W, alpha = eng.get_factor()
    K = O.kernel_matrix(kind, 1.0, ls, X, X) + noise * np.eye(N)
    Lref = np.linalg.cholesky(K)
    assert_close(L, Lref, rtol=1e-9, atol=1e-9 * np.abs(Lref).max(), what="L at N = 8200")
    R = np.tril(W) @ np.tril(L)
    assert np.abs(R - np.eye(N)).max() < 1e-8
    assert_close(alpha, np.linalg.solve(Lref.T, np.linalg.solve(Lref, Y - c)), rtol=1e-6,
                 atol=1e-6 * np.abs(alpha).max(), what="alpha at N = 8200")
    eng.set_data(X, Y)
    L2, W2, a2 = eng.get_factor()
    assert np.array_equal(L2, L) and np.array_equal(W2, W) and np.array_equal(a2, alpha)


@pytest.mark.parametrize("N", [640, 1536])
def test_dag_update_many_fresh_handles(N):
    """Forty fresh handles in recycled device memory, every one a first update: the persistent kernel must not depend
    on how its workgroups and waves happen to be timed (a race of the leaf's two panel waves, exposed by a faster
    dispatcher, corrupted ~15 % of such updates before it was fixed)."""
    X, Y, ls, c, kind, noise = _problem(N)
    ref = _engine(X, Y, ls, c, kind, noise)
    Lr, Wr, ar = ref.get_factor()
    for _ in range(40):
        e = _engine(X, Y, ls, c, kind, noise)
        L, W, al = e.get_factor()
        assert np.array_equal(L, Lr) and np.array_equal(W, Wr) and np.array_equal(al, ar)
        e.close()


def test_dag_update_two_handles_concurrently_on_private_streams():
    """Two persistent launches share the GPU (find_best_model_initialization drives several engines from several
    threads): neither may depend on being fully resident."""
    import threading

    X, Y, ls, c, kind, noise = _problem(2048, d=6)
    ref = _engine(X, Y, ls, c, kind, noise)
    Lr, Wr, _ = ref.get_factor()
    engs = [_engine(X, Y, ls, c, kind, noise) for _ in range(2)]
    for e in engs:
        e.use_private_stream()
    errs = []

    def work(e):
        try:
            for _ in range(5):
                e.set_data(X, Y)
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=work, args=(e,)) for e in engs]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    for e in engs:
        L, W, _ = e.get_factor()
        assert np.array_equal(L, Lr) and np.array_equal(W, Wr)


def test_update_concurrency_shares_the_gpu_and_keeps_the_bits():
    """tgp_set_update_concurrency: the persistent kernel on 1 / n of the compute units (another dispatch order, fewer
    workers) gives the same factor bit for bit, alone and with n handles factorising side by side."""
    import threading

    X, Y, ls, c, kind, noise = _problem(2048, d=6)
    ref = _engine(X, Y, ls, c, kind, noise)
    Lr, Wr, ar = ref.get_factor()
    for n in (2, 8, 16):
        e = _engine(X, Y, ls, c, kind, noise)
        e.set_update_concurrency(n)
        e.set_data(X, Y)
        L, W, al = e.get_factor()
        assert np.array_equal(L, Lr) and np.array_equal(W, Wr) and np.array_equal(al, ar), n
        e.close()
    with pytest.raises(ValueError):
        ref.set_update_concurrency(0)
    engs = [_engine(X, Y, ls, c, kind, noise) for _ in range(4)]
    for e in engs:
        e.use_private_stream()
        e.set_update_concurrency(4)
    errs = []

    def work(e):
        try:
            for _ in range(6):
                e.set_data(X, Y)
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=work, args=(e,)) for e in engs]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    for e in engs:
        L, W, _ = e.get_factor()
        assert np.array_equal(L, Lr) and np.array_equal(W, Wr)


@pytest.mark.parametrize("N,noise", [(N, s2) for N in (300, 384, 640, 1100, 2049, 4096) for s2 in (1e-2, 1e-5)] + [(5000, 1e-2)])
def test_two_workgroup_chain_against_the_one_workgroup_chain(N, noise):
    """Round 6: the chain of the persistent kernel as TWO workgroups swapping roles (leaf / helper: csrc/tgp_kernels_dag.hip
    run_duo) -- the default wherever the split plan applies.  L(j+1,j) is a blocked triangular solve against L_jj there and a
    product with W_jj in the one-workgroup chain (variant bit 9): the same factor and inverse up to rounding, NOT the same bits;
    bit-identical run to run and across handles like every other form; the residuals |W L - I| and |L L^T - K| at the level of the
    one-workgroup chain's."""
    X, Y, ls, c, kind, _ = _problem(N, d=4 if N < 4096 else 8, noise=noise)
    duo = _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL)
    one = _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL | DAG_ONE_CHAIN)
    Ld, Wd, ad = duo.get_factor()
    Lo, Wo, ao = one.get_factor()
    cond = 1.0 + N / noise
    tol = 64 * np.finfo(float).eps * cond
    assert_close(Ld, Lo, rtol=1e-9, atol=tol * np.abs(Lo).max(), what="L: two workgroups vs one")
    assert_close(Wd, Wo, rtol=1e-7, atol=tol * np.abs(Wo).max() * 64, what="W: two workgroups vs one")
    assert np.array_equal(np.triu(Ld, 1), np.zeros_like(Ld)) and np.array_equal(np.triu(Wd, 1), np.zeros_like(Wd))
    st = O.gpr_update(kind, 1.0, ls, noise, c, X, Y)
    K = st.L @ st.L.T
    rd, ro = np.abs(Ld @ Ld.T - K).max(), np.abs(Lo @ Lo.T - K).max()
    assert rd <= 4 * ro + 64 * np.finfo(float).eps * np.abs(K).max(), (rd, ro)
    wd, wo = np.abs(np.tril(Wd) @ np.tril(Ld) - np.eye(N)).max(), np.abs(np.tril(Wo) @ np.tril(Lo) - np.eye(N)).max()
    assert wd <= 4 * wo + 1e-12, (wd, wo)
    assert_close(ad, ao, rtol=1e-6, atol=1e-6 * np.abs(ao).max(), what="alpha")
    for _ in range(2):
        duo.set_data(X, Y)
        L2, W2, a2 = duo.get_factor()
        assert np.array_equal(L2, Ld) and np.array_equal(W2, Wd) and np.array_equal(a2, ad)
    other = _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL)
    L3, W3, a3 = other.get_factor()
    assert np.array_equal(L3, Ld) and np.array_equal(W3, Wd) and np.array_equal(a3, ad)


def test_default_policy_is_the_persistent_kernel_from_npad_512_on():
    """Round 6: `update` is the persistent launch from Npad = 512 on (rounds 3 - 5: 4096) -- measured faster than the recursion from
    N ~ 640 on and, above all, the size rule the batched prior draws of the fit hang on (profiles/r06_dag_small_sizes*.txt).  The
    default policy gives the bits of the forced persistent kernel (variant bit 5) there, and the recursion's below."""
    for N, persistent in ((200, False), (256, False), (300, True), (600, True), (1100, True)):
        X, Y, ls, c, kind, noise = _problem(N, d=4)
        default = _engine(X, Y, ls, c, kind, noise, variant=0)
        assert default.update_is_persistent(N) == persistent, N
        other = _engine(X, Y, ls, c, kind, noise, variant=DAG_SMALL if persistent else NO_DAG)
        for a, b in zip(default.get_factor(), other.get_factor()):
            assert np.array_equal(a, b), N


@pytest.mark.parametrize("N,variant", [(200, 0), (300, 0), (640, DAG_SMALL), (2049, DAG_SMALL), (4096, 0)])
def test_trial_evaluation_matches_the_likelihood_of_a_full_update(N, variant):
    """tgp_nlml_trial (factor-only launch + block forward substitution where the persistent kernel applies, a full update
    below): the same value as set_data + nlml(value only) and as the oracle, at several hyper-parameter draws on the data
    already on the device; no posterior is left behind; a later set_data gives the usual factor bit for bit."""
    from trieste_amd._lib import TgpError

    X, Y, ls, c, kind, noise = _problem(N, d=5)
    ref = _engine(X, Y, ls, c, kind, noise, variant=variant)
    Lr, Wr, ar = ref.get_factor()
    eng = _engine(X, Y, ls, c, kind, noise, variant=variant)
    rng = np.random.default_rng(N)
    for trial in range(4):
        ls_t = ls * np.exp(0.3 * rng.standard_normal(len(ls)))
        var_t, noise_t, c_t = float(np.exp(0.2 * rng.standard_normal())), noise * (1.0 + trial), c + 0.1 * trial
        ref.set_hyper(var_t, ls_t, noise_t, c_t)
        ref.set_data(X, Y)
        want = ref.nlml(False)[0]
        eng.set_hyper(var_t, ls_t, noise_t, c_t)
        got = eng.nlml_trial()
        assert abs(got - want) <= 1e-9 * abs(want) + 1e-9 * N, (trial, got, want)
        st = O.gpr_update(kind, var_t, ls_t, noise_t, c_t, X, Y)
        assert abs(got - O.nlml_and_grad(st)[0]) <= 1e-8 * abs(want) + 1e-8 * N
        assert eng.nlml_trial() == got                      # the same bits again
    if eng.update_is_persistent(N):                        # factor only: nothing to query afterwards (Npad >= 512 since round 6)
        with pytest.raises(RuntimeError):                   # TGP_ERR_STATE
            eng.predict(X[:3])
    eng.set_hyper(1.0, ls, noise, c)
    eng.set_data(X, Y)
    L, W, al = eng.get_factor()
    assert np.array_equal(L, Lr) and np.array_equal(W, Wr) and np.array_equal(al, ar)
    fresh = _engine(X, Y, ls, c, kind, noise, variant=variant)
    fresh.close()


def test_trial_evaluation_needs_data_and_reports_a_breakdown():
    from trieste_amd._lib import NotPositiveDefiniteError, TgpError
    from trieste_amd.engine import GPEngine

    eng = GPEngine(2, "rbf")
    eng.set_variant(DAG_SMALL)
    eng.set_hyper(1.0, [0.3, 0.3], 1e-2, 0.0)
    with pytest.raises(RuntimeError):                       # TGP_ERR_STATE: nothing uploaded yet
        eng.nlml_trial()
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(1024, 2))
    X[600:] = 0.5 + 1e-13 * rng.standard_normal((424, 2))
    Y = rng.standard_normal(1024)
    eng.set_data(X, Y)
    eng.set_hyper(1.0, [0.3, 0.3], 1e-300, 0.0)
    with pytest.raises(NotPositiveDefiniteError):
        eng.nlml_trial()
    eng.set_hyper(1.0, [0.3, 0.3], 1e-2, 0.0)
    assert np.isfinite(eng.nlml_trial())


@pytest.mark.parametrize("where", ["first_block", "late_blocks"])
def test_dag_update_reports_a_matrix_that_is_not_positive_definite(where):
    """A breakdown fills the rest of the factor with NaN; every task still runs and raises its flag (no hang), the
    breakdown is reported, and the handle is usable afterwards."""
    from trieste_amd._lib import NotPositiveDefiniteError
    from trieste_amd.engine import GPEngine

    rng = np.random.default_rng(0)
    N, d = 1024, 2
    X = rng.uniform(size=(N, d))
    if where == "first_block":
        X[1] = X[0]                                   # pivot 1 is exactly zero
    else:
        X[600:] = 0.5 + 1e-13 * rng.standard_normal((N - 600, d))   # a numerically rank-one trailing block
    Y = rng.standard_normal(N)
    eng = GPEngine(d, "rbf")
    eng.set_variant(DAG_SMALL)
    eng.set_hyper(1.0, [0.3, 0.3], 1e-300, 0.0)      # (numerically) no noise
    with pytest.raises(NotPositiveDefiniteError):
        eng.set_data(X, Y)
    eng.set_hyper(1.0, [0.3, 0.3], 1e-2, 0.0)         # the handle is usable afterwards
    eng.set_data(X, Y)
    L, W, _ = eng.get_factor()
    assert np.isfinite(L).all() and np.abs(np.tril(W) @ np.tril(L) - np.eye(N)).max() < 1e-8


def test_batched_trial_evaluations_equal_the_single_ones_bit_for_bit():
    """tgp_nlml_trial_batch at N = 4096: 11 members in ONE persistent launch, then 19 (a launch of ten and one of nine)
    -- B chain workgroups, one task list over all members.  Every value equals tgp_nlml_trial's at the same
    hyper-parameters bit for bit; a member whose kernel matrix is not positive definite gets NaN / not-ok and does not
    disturb the others; the engine's own hyper-parameters and posterior are untouched."""
    N, d = 4096, 5
    X, Y, ls, c, kind, noise = _problem(N, d=d)
    X = X.copy()
    X[3000:3040] = X[2999] + 1e-13 * np.random.default_rng(1).standard_normal((40, d))  # near-duplicates: PD only with noise
    eng = _engine(X, Y, ls, c, kind, noise)
    before = eng.predict(X[:50] + 0.01)
    rng = np.random.default_rng(4096)
    hy = []
    for b in range(11):
        ls_t = ls * np.exp(0.3 * rng.standard_normal(d))
        hy.append(np.concatenate([[float(np.exp(0.2 * rng.standard_normal()))], ls_t, [noise * (1.0 + b), c + 0.1 * b]]))
    hy = np.array(hy)
    hy[4, 1 + d] = 1e-300                                   # member 4: (numerically) no noise -> not positive definite
    values, ok = eng.nlml_trial_batch(hy)
    assert ok.tolist() == [b != 4 for b in range(11)] and np.isnan(values[4]) and np.isfinite(np.delete(values, 4)).all()
    again, ok2 = eng.nlml_trial_batch(hy)
    np.testing.assert_array_equal(np.delete(again, 4), np.delete(values, 4))      # the same bits again
    after = eng.predict(X[:50] + 0.01)
    np.testing.assert_array_equal(before[0], after[0])                            # the posterior was not touched
    np.testing.assert_array_equal(before[1], after[1])
    single = _engine(X, Y, ls, c, kind, noise)
    for b in (0, 3, 5, 7, 8, 10):                            # members of both launches against the one-by-one evaluation
        single.set_hyper(hy[b, 0], hy[b, 1:1 + d], hy[b, 1 + d], hy[b, 2 + d])
        assert single.nlml_trial() == values[b], (b, single.nlml_trial(), values[b])
    st = O.gpr_update(kind, hy[0, 0], hy[0, 1:1 + d], hy[0, 1 + d], hy[0, 2 + d], X, Y)
    assert abs(values[0] - O.nlml_and_grad(st)[0]) <= 1e-8 * abs(values[0]) + 1e-8 * N
    # more members than one launch takes (16): two launches, enqueued back to back; the first eleven are the same bits
    hy19 = np.concatenate([hy, hy[[0, 1, 2, 3, 5, 6, 7, 8]] * (1.0 + 1e-3)])
    v19, ok19 = eng.nlml_trial_batch(hy19)
    np.testing.assert_array_equal(np.delete(v19[:11], 4), np.delete(values, 4))
    assert ok19[11:].all() and np.isfinite(v19[11:]).all() and not ok19[4]
    # a permutation of the members permutes the values (members do not interact)
    perm = rng.permutation(11)
    pv, _ = eng.nlml_trial_batch(hy[perm])
    np.testing.assert_array_equal(np.delete(pv, int(np.where(perm == 4)[0][0])), np.delete(values[perm], int(np.where(perm == 4)[0][0])))
    # the process-wide scratch (3 N^2 doubles per member) can be handed back; the next call allocates it again: same bits
    import torch
    free0 = torch.cuda.mem_get_info()[0]
    eng.release_scratch()
    assert torch.cuda.mem_get_info()[0] >= free0 + 10 * 3 * 4096 * 4096 * 8   # (the launch of ten members' matrices, at least)
    again, _ = eng.nlml_trial_batch(hy)
    np.testing.assert_array_equal(np.delete(again, 4), np.delete(values, 4))


def test_batched_trial_evaluations_below_the_persistent_size():
    """Below N = 257 (Npad = 256: the recursion of dependent launches; round 5: below 3841) the members run one after the other on the
    handle itself: tgp_nlml_trial's values, the handle's hyper-parameters and posterior restored."""
    X, Y, ls, c, kind, noise = _problem(200, d=4)
    eng = _engine(X, Y, ls, c, kind, noise, variant=0)
    before = eng.predict(X[:20] + 0.01)
    rng = np.random.default_rng(3)
    hy = np.array([np.concatenate([[1.0 + 0.1 * b], ls * np.exp(0.2 * rng.standard_normal(4)), [noise, c]]) for b in range(5)])
    values, ok = eng.nlml_trial_batch(hy)
    assert ok.all()
    single = _engine(X, Y, ls, c, kind, noise, variant=0)
    for b in range(5):
        single.set_hyper(hy[b, 0], hy[b, 1:5], hy[b, 5], hy[b, 6])
        assert single.nlml_trial() == values[b]
    after = eng.predict(X[:20] + 0.01)
    np.testing.assert_array_equal(before[0], after[0])
    np.testing.assert_array_equal(before[1], after[1])
    with pytest.raises(ValueError):
        eng.nlml_trial_batch(np.ones((2, 3)))
