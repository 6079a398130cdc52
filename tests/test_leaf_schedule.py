"""The 128-leaf's schedule (trieste_amd/csrc/tgp_kernels_leaf.hip), restated on numpy blocks.

The kernel factors a 128 x 128 block panel by panel and builds W = L^-1 in place, right-looking, with most of
the work deferred to "worker" waves that run concurrently with the next panel.  What makes that legal is a set
of claims about which LDS slot holds what, when, and who may still read it.  This test executes the same item
lists -- panel, worker items [P kb], column / row items [C kb] -- on one shared array, with the items of each
phase in RANDOM order (any order must give the same result if the phase is hazard free, which is what lets
the waves of a phase run unsynchronised), and checks L and W against numpy.  It pins the algebra and the
hazard analysis of the schedule on the CPU; the kernel itself is checked by the `-m gpu` factor tests.
"""
import numpy as np
import pytest

QB, B = 8, 16  # 16-blocks per side, block size


def _blk(S, bi, bj):
    return S[B * bi:B * (bi + 1), B * bj:B * (bj + 1)]


def _panel(S, L, kb):
    """Panel kb: factor the diagonal block, solve the rows below it, leave W_d in slot (kb, kb) and L[bi][kb]
    in the slots below (panel_factor: rows below ride on the same eliminations, identity rows give W_d)."""
    r0 = B * kb
    Ld = np.linalg.cholesky(np.tril(S[r0:r0 + B, r0:r0 + B]) + np.tril(S[r0:r0 + B, r0:r0 + B], -1).T)
    Wd = np.linalg.inv(Ld)
    below = S[r0 + B:, r0:r0 + B] @ Wd.T
    L[r0:r0 + B, r0:r0 + B] = Ld
    L[r0 + B:, r0:r0 + B] = below
    S[r0:r0 + B, r0:r0 + B] = Wd
    S[r0 + B:, r0:r0 + B] = below


def _worker_items(kb):
    """make_work_item(kb, idx) for every idx: (a1, b1, b1 transposed?, out, a2, b2, keep)."""
    pb = kb - 1
    items = []
    for bi in range(kb + 1, QB):  # trailing updates of panel pb for block columns > kb
        for bj in range(kb + 1, bi + 1):
            items.append(((bi, pb), (bj, pb), True, (bi, bj), None, None, 1.0))
    if pb >= 1:  # rows bi >= kb of T against source row pb
        for bi in range(kb, QB):
            for c in range(pb):
                if c < pb - 1:
                    items.append(((bi, pb), (pb, c), False, (bi, c), None, None, 1.0))
                else:  # fused with the initial term that replaces the L block
                    items.append(((bi, pb), (pb, c), False, (bi, c), (bi, pb - 1), (pb - 1, pb - 1), 0.0))
    items.append(((kb, pb), (pb, pb), False, "scratch", None, None, 0.0))  # T~[kb][pb] for [C kb]
    return items


def _run_items(S, scratch, items, rng):
    """out = keep * out - (A1 B1 [+ A2 B2]); every item reads all its operands before it writes (one wave)."""
    for k in rng.permutation(len(items)):
        a1, b1, bt, o, a2, b2, keep = items[k]
        B1 = _blk(S, *b1)
        p = _blk(S, *a1) @ (B1.T if bt else B1)
        if a2 is not None:
            p = p + _blk(S, *a2) @ _blk(S, *b2)
        if o == "scratch":
            scratch[...] = -p
        else:
            out = _blk(S, *o)
            out[...] = keep * out - p


def _c_phase(S, scratch, kb, rng):
    """[C kb]: column kb + 1 of the trailing update, and row kb of T."""
    pb = kb - 1
    jobs = [("col", bi) for bi in range(kb + 1, QB)] + [("row", c) for c in range(kb)]
    for k in rng.permutation(len(jobs)):
        kind, x = jobs[k]
        if kind == "col":
            out = _blk(S, x, kb + 1)
            out[...] = out - _blk(S, x, kb) @ _blk(S, kb + 1, kb).T
        else:
            src = _blk(S, kb, x) if x < pb else scratch
            _blk(S, kb, x)[...] = _blk(S, kb, kb) @ src


@pytest.mark.parametrize("seed", range(6))
def test_in_place_schedule_gives_L_and_its_inverse_whatever_the_order_inside_a_phase(seed):
    rng = np.random.default_rng(seed)
    n = QB * B
    G = rng.standard_normal((n, n + 8))
    A = G @ G.T + 0.5 * np.eye(n)
    S = np.tril(A).copy()  # the kernel loads the lower triangle, zeros above
    L = np.zeros_like(S)
    scratch = np.zeros((B, B))
    for kb in range(QB):
        # [P kb]: the panel (block column kb only) and, concurrently, the worker items (never block column kb)
        if rng.integers(2):  # either side may finish first
            _panel(S, L, kb)
            if kb >= 1:
                _run_items(S, scratch, _worker_items(kb), rng)
        else:
            if kb >= 1:
                _run_items(S, scratch, _worker_items(kb), rng)
            _panel(S, L, kb)
        _c_phase(S, scratch, kb, rng)
    Lref = np.linalg.cholesky(A)
    np.testing.assert_allclose(L, Lref, rtol=0, atol=1e-10 * np.abs(Lref).max())
    W = np.tril(S)
    np.testing.assert_allclose(W @ Lref, np.eye(n), rtol=0, atol=1e-8)


def test_item_counts_fit_the_worker_slots():
    """At most 22 items per panel for six workers with four slots each (NWORK x WSLOTS in the kernel)."""
    for kb in range(1, QB):
        assert len(_worker_items(kb)) <= 24
        pb = kb - 1
        ns = (QB - 1 - kb) * (QB - kb) // 2
        nt = (QB - kb) * pb
        assert len(_worker_items(kb)) == ns + nt + 1


def test_no_item_of_a_phase_writes_what_another_one_touches():
    """Static form of the hazard analysis: inside [P kb] (the panel's block column kb included) and inside [C kb],
    the block an item writes is neither read nor written by any other item of the phase."""
    for kb in range(QB):
        pb = kb - 1
        phase_p = []  # (reads, write)
        col_kb = {(bi, kb) for bi in range(kb, QB)}
        phase_p.append((set(col_kb), None))  # the panel reads and writes block column kb: treated as one unit
        panel_writes = col_kb
        if kb >= 1:
            for a1, b1, _bt, o, a2, b2, keep in _worker_items(kb):
                reads = {a1, b1} | ({a2, b2} if a2 is not None else set()) | ({o} if keep else set())
                phase_p.append((reads, o))
        for i, (ri, wi) in enumerate(phase_p):
            writes_i = panel_writes if wi is None else {wi}
            for j, (rj, wj) in enumerate(phase_p):
                if i == j:
                    continue
                writes_j = panel_writes if wj is None else {wj}
                assert not (writes_i & (rj | writes_j)), (kb, i, j, writes_i & (rj | writes_j))
        phase_c = []
        for bi in range(kb + 1, QB):
            phase_c.append(({(bi, kb), (kb + 1, kb), (bi, kb + 1)}, (bi, kb + 1)))
        for c in range(kb):
            phase_c.append(({(kb, kb), (kb, c) if c < pb else "scratch"}, (kb, c)))
        for i, (ri, wi) in enumerate(phase_c):
            for j, (rj, wj) in enumerate(phase_c):
                if i != j:
                    assert wi != wj and wi not in rj, (kb, i, j)
