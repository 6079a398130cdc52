"""The static task list of the persistent `update` kernel (csrc/tgp_kernels_dag.hip, exported for tests through
``tgp_dag_plan``), checked WITHOUT a GPU:

* executed on numpy blocks -- chain steps and tile tasks interpreted exactly as the kernel defines them -- it produces
  L = chol(A) and W = L^-1, with the kernel's dispatch rule (workers draw the positions of ONE list in order and wait
  for the flags of what they drew; the chain is its own worker) AND in random valid interleavings;
* the dispatch order is a topological order -- every flag a task waits for belongs to a chain step or to a task EARLIER
  in the list -- which makes the in-order dispatch deadlock-free whatever the residency; the dependency graph as a
  whole (tasks + chain steps) is acyclic;
* every pair of accesses to the same tile with a write among them is ordered by the flags (happens-before through the
  transitive closure): no data race, hence a schedule-independent -- bit-identical -- result.
Reference: the factorisation behind trieste/models/gpflow/models.py:171-186 -> interface.py:108-112."""
import ctypes as C

import numpy as np
import pytest

from trieste_amd import _lib

T = 128
NONE = 0xFFFFFFFF
NN, BETA, NEG, HALF, HI = 1, 2, 4, 8, 16
SPLIT = 2   # tgp_dag_plan flags bit 1: the split plan of the single full update (round 6)
SIB = 32    # DagTask flag: dep3 is the lower-half sibling (the two-workgroup chain's plan splits every T and every last product)
DUO = 4     # bit 2: the order (and worker count) of the launch whose chain is two workgroups (the same tasks and flags)


class Task(C.Structure):
    _fields_ = [("a_off", C.c_uint32), ("b_off", C.c_uint32), ("c_off", C.c_uint32), ("o_off", C.c_uint32),
                ("nk", C.c_uint32), ("flags", C.c_uint32), ("a_mat", C.c_uint8), ("b_mat", C.c_uint8),
                ("c_mat", C.c_uint8), ("o_mat", C.c_uint8), ("dep", C.c_uint32 * 3), ("set", C.c_uint32),
                ("dep3", C.c_uint32)]

    @property
    def deps(self):   # the flags the task waits for before it STARTS: four slots, three for an upper half with a sibling (SIB)
        return list(self.dep) + ([] if self.flags & SIB else [self.dep3])

    @property
    def sibling(self):   # SIB: the lower half whose flag this task waits for at its END, before its own flag goes up
        return self.dep3 if self.flags & SIB else NONE

    @property
    def rows(self):   # the rows of the output tile (and of the A operand) the task computes
        if not self.flags & HALF:
            return slice(0, T)
        return slice(T // 2, T) if self.flags & HI else slice(0, T // 2)


def plan(nb, ld=None, flags=0):
    lib = _lib.load()
    ld = ld or nb * T
    n, nu = C.c_int64(), C.c_int64()
    rc = lib.tgp_dag_plan(nb, ld, None, 0, C.byref(n), C.byref(nu), None, None, flags)
    assert rc == _lib.TGP_ERR_SHAPE and n.value >= 0
    tasks = (Task * max(n.value, 1))()
    chain = (C.c_uint32 * (3 * nb))()   # [2 j] diagonal step, [2 j + 1] sub-diagonal step, [2 nb + j] its second flag (split plan)
    order = (C.c_uint32 * max(n.value, 1))()
    assert lib.tgp_dag_plan(nb, ld, tasks, n.value, C.byref(n), C.byref(nu), chain, order, flags) == _lib.TGP_OK
    assert 0 <= nu.value <= n.value
    ORDER[(nb, ld)] = [order[i] for i in range(n.value)]
    return [tasks[i] for i in range(n.value)], list(chain), ld, nu.value


ORDER = {}  # (nb, ld) -> dispatch order of the last plan() call


def tile_of(off, ld):
    r, c = divmod(off, ld)
    assert r % T == 0 and c % T == 0
    return r // T, c // T


def both(mat, i, j):
    """a whole tile as its two row halves (the unit a half-tile task of the split plan reads and writes)"""
    return {(mat, i, j, 0), (mat, i, j, 1)}


def accesses(t, ld):
    """(reads, writes) of a bulk task as sets of (matrix, tile row, tile col, row half)."""
    reads, writes = set(), set()
    ai, ak = tile_of(t.a_off, ld)
    bi, bk = tile_of(t.b_off, ld)
    mine = [1 if t.flags & HI else 0] if t.flags & HALF else [0, 1]   # the row halves of the A operand, of C and of the output
    for kt in range(t.nk):
        reads |= {(t.a_mat, ai, ak + kt, h) for h in mine}
        reads |= both(t.b_mat, bi + kt, bk) if t.flags & NN else both(t.b_mat, bi, bk + kt)
    if t.flags & BETA:
        reads |= {(t.c_mat,) + tile_of(t.c_off, ld) + (h,) for h in mine}
    writes |= {(t.o_mat,) + tile_of(t.o_off, ld) + (h,) for h in mine}
    return reads, writes


def chain_accesses(nb):
    """per chain node (A(j) = diagonal step, B(j) = sub-diagonal step): reads, writes, flag it sets (relative)"""
    out = []
    for j in range(nb):
        out.append((both(0, j, j), both(1, j, j) | both(2, j, j)))              # P(j,j) -> L_jj, W_jj (Lsub stays in LDS)
        out.append((both(0, j + 1, j) | both(2, j, j) if j + 1 < nb else set(), both(1, j + 1, j) if j + 1 < nb else set()))
    return out


class Machine:
    """numpy interpretation of the kernel's task semantics on an ld x ld workspace."""

    def __init__(self, A, nb, tasks, chain, ld, nu):
        self.m = [A.copy(), np.zeros_like(A), np.zeros_like(A)]
        self.nb, self.tasks, self.chain, self.ld, self.nu = nb, tasks, chain, ld, nu
        self.flags = np.zeros(len(tasks) + 2 * nb, dtype=bool)
        self.taken = np.zeros(len(tasks), dtype=bool)
        self.order = ORDER[(nb, ld)]
        self.pos = 0
        self.chain_pos = 0  # 2 j (diagonal step of j) or 2 j + 1
        self.lsub = None
        self.parked = {}    # lower half -> an upper half that has finished and waits for it (SIB)

    def blk(self, mat, i, j):
        return self.m[mat][i * T:(i + 1) * T, j * T:(j + 1) * T]

    def ready(self, t):
        return all(d == NONE or self.flags[d] for d in t.deps)

    def run_bulk(self, idx):
        t, ld = self.tasks[idx], self.ld
        assert self.ready(t), f"task {idx} started before its flags"
        ai, ak = tile_of(t.a_off, ld)
        bi, bk = tile_of(t.b_off, ld)
        rows = t.rows   # (a half-tile task: the same products restricted to its rows of A, C and the output)
        acc = np.zeros((T, T))[rows]
        for kt in range(t.nk):
            a = self.blk(t.a_mat, ai, ak + kt)[rows]
            acc += a @ self.blk(t.b_mat, bi + kt, bk) if t.flags & NN else a @ self.blk(t.b_mat, bi, bk + kt).T
        cin = self.blk(t.c_mat, *tile_of(t.c_off, ld))[rows].copy() if t.flags & BETA else 0.0
        oi, oj = tile_of(t.o_off, ld)
        self.m[t.o_mat][oi * T:(oi + 1) * T, oj * T:(oj + 1) * T][rows] = cin + (-acc if t.flags & NEG else acc)
        assert t.set == idx
        if t.sibling != NONE and not self.flags[t.sibling]:
            self.parked[t.sibling] = idx      # the kernel's worker polls the sibling's flag here; its own flag stays down
            return
        self.flags[idx] = True
        if idx in self.parked:
            self.flags[self.parked.pop(idx)] = True

    def chain_ready(self):
        if self.chain_pos >= 2 * self.nb:
            return False
        need = [self.chain[self.chain_pos]]
        if self.chain_pos % 2 == 1:                       # the sub-diagonal step: a second flag in the split plan
            need.append(self.chain[2 * self.nb + self.chain_pos // 2])
        return all(d == NONE or self.flags[d] for d in need)

    def run_chain(self):
        j, part = divmod(self.chain_pos, 2)
        nt = len(self.tasks)
        if part == 0:
            S = self.blk(0, j, j).copy()
            if j > 0:
                S -= self.lsub @ self.lsub.T
            Ljj = np.linalg.cholesky(np.tril(S) + np.tril(S, -1).T)
            self.m[1][j * T:(j + 1) * T, j * T:(j + 1) * T] = Ljj
            self.m[2][j * T:(j + 1) * T, j * T:(j + 1) * T] = np.tril(np.linalg.inv(Ljj))  # (the leaf writes zeros above)
            self.flags[nt + j] = True
        elif j + 1 < self.nb:
            assert self.flags[nt + j]
            self.lsub = self.blk(0, j + 1, j) @ self.blk(2, j, j).T
            self.m[1][(j + 1) * T:(j + 2) * T, j * T:(j + 1) * T] = self.lsub
            self.flags[nt + self.nb + j] = True
        self.chain_pos += 1

    def acquire(self):
        """the kernel's rule with ONE worker: the next position of the dispatch order, if its flags are up"""
        if self.pos < len(self.order):
            i = self.order[self.pos]
            if self.ready(self.tasks[i]):
                self.pos += 1
                self.taken[i] = True
                return i
        return None

    def acquire_any(self):
        """a worker that drew some later position: any ready task not yet taken"""
        for i in self.order:
            if not self.taken[i] and self.ready(self.tasks[i]):
                self.taken[i] = True
                return i
        return None

    def lists_done(self):
        return bool(self.taken.all())

    def done(self):
        return self.lists_done() and self.chain_pos >= 2 * self.nb


def spd(n, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, 3))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d2 / 0.3 ** 2) + 1e-2 * np.eye(n)


@pytest.mark.parametrize("split", [0, SPLIT, SPLIT | DUO], ids=["whole-tiles", "split", "split-two-workgroup-chain"])
@pytest.mark.parametrize("nb", [1, 2, 4, 7, 49])  # (49: bursts of 8 k tiles)
def test_plan_in_list_order_factors_and_inverts(nb, split):
    tasks, chain, ld, nu = plan(nb, flags=split)
    if split and nb >= 3:   # T(i, i-2) and the last burst of tile (i, i-1), i = 2 .. nb - 1, as two half-tile tasks each
        halves = [t for t in tasks if t.flags & HALF]
        if split & DUO:     # ... the two-workgroup chain's plan: EVERY T(i,j) and the last burst of EVERY tile below the diagonal,
            #                 every upper half publishing for both (SIB: its dep3 is the lower half, an earlier task)
            assert len(halves) == 2 * (nb - 1) * (nb - 2) and sum(1 for t in halves if t.flags & HI) == (nb - 1) * (nb - 2)
            assert all(bool(t.flags & SIB) == bool(t.flags & HI) for t in halves)
            assert all(tasks[t.dep3].flags & HALF and not tasks[t.dep3].flags & HI and tasks[t.dep3].o_off == t.o_off
                       for t in halves if t.flags & SIB)
        else:
            assert len(halves) == 4 * (nb - 2) and sum(1 for t in halves if t.flags & HI) == 2 * (nb - 2)
            assert not any(t.flags & SIB for t in tasks)
        assert all(t.nk == 1 for t in halves)
        assert sum(1 for j in range(nb) if chain[2 * nb + j] != NONE) == nb - 2
    else:
        assert not any(t.flags & HALF for t in tasks) and all(c == NONE for c in chain[2 * nb:])
    n = nb * T
    A = spd(n, nb)
    mc = Machine(A, nb, tasks, chain, ld, nu)
    while not mc.done():  # one worker with the kernel's dispatch rule; the chain runs whenever it can
        if mc.chain_ready():
            mc.run_chain()
            continue
        i = mc.acquire()
        assert i is not None, f"stuck: chain at {mc.chain_pos}, {int(mc.taken.sum())} of {len(tasks)} tasks taken"
        mc.run_bulk(i)
    L = np.tril(mc.m[1])
    np.testing.assert_allclose(L @ L.T, A, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(np.tril(mc.m[2]) @ L, np.eye(n), atol=1e-9)
    np.testing.assert_array_equal(np.triu(mc.m[1], 1), 0.0)   # nothing is ever written above the diagonal
    np.testing.assert_array_equal(np.triu(mc.m[2], 1), 0.0)


@pytest.mark.parametrize("nb", [2, 5, 9])
def test_factor_only_plan_builds_the_factor_and_the_diagonal_inverses(nb):
    """The plan of tgp_nlml_trial (flags bit 0): no task of the inverse -- about half the products -- L complete, W
    holding exactly its diagonal tiles W_jj = L_jj^-1 (what the T tasks and the block forward substitution need)."""
    full, _, _, _ = plan(nb)
    tasks, chain, ld, nu = plan(nb, flags=1)
    assert len(tasks) < len(full) and all(t.o_mat != 2 for t in tasks)   # 2 = DAG_MAT_W: nobody writes W but the chain
    n = nb * T
    A = spd(n, 40 + nb)
    mc = Machine(A, nb, tasks, chain, ld, nu)
    while not mc.done():
        if mc.chain_ready():
            mc.run_chain()
            continue
        i = mc.acquire()
        assert i is not None
        mc.run_bulk(i)
    L = np.tril(mc.m[1])
    np.testing.assert_allclose(L @ L.T, A, rtol=1e-11, atol=1e-11)
    W = mc.m[2]
    for j in range(nb):
        blk = slice(j * T, (j + 1) * T)
        np.testing.assert_allclose(np.tril(W[blk, blk]) @ L[blk, blk], np.eye(T), atol=1e-9)
        W[blk, blk] = 0.0
    np.testing.assert_array_equal(W, 0.0)
    # the block forward substitution the trial evaluation runs on it: z = L^-1 r from L and the W_jj alone
    r = np.random.default_rng(nb).standard_normal(n)
    Wd = mc.m[2]
    z = np.zeros(n)
    for i in range(nb):
        bi = slice(i * T, (i + 1) * T)
        acc = r[bi] - L[bi, : i * T] @ z[: i * T]
        z[bi] = np.tril(np.linalg.inv(L[bi, bi])) @ acc
    np.testing.assert_allclose(L @ z, r, atol=1e-9)


@pytest.mark.parametrize("split", [0, SPLIT, SPLIT | DUO], ids=["whole-tiles", "split", "split-two-workgroup-chain"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_plan_in_random_valid_interleavings_gives_the_same_bits(seed, split):
    """Three workers with a window: any ready task among the next few popped ones may run, in any order against the
    chain -- what different residencies / timings produce on the device.  Same L and W, bit for bit."""
    nb = 5
    tasks, chain, ld, nu = plan(nb, flags=split)
    A = spd(nb * T, 11)
    ref = Machine(A, nb, tasks, chain, ld, nu)
    while not ref.done():
        if ref.chain_ready():
            ref.run_chain()
        else:
            ref.run_bulk(ref.acquire())
    rng = np.random.default_rng(seed)
    mc = Machine(A, nb, tasks, chain, ld, nu)
    popped = []  # tasks taken by workers (their flags were up) but not yet run
    while not mc.done() or popped:
        while len(popped) < 3:
            i = mc.acquire_any()
            if i is None:
                break
            popped.append(i)
        choices = [("b", i) for i in popped] + ([("c", -1)] if mc.chain_ready() else [])
        assert choices, "deadlock"
        kind, i = choices[rng.integers(len(choices))]
        if kind == "c":
            mc.run_chain()
        else:
            mc.run_bulk(i)
            popped.remove(i)
    np.testing.assert_array_equal(mc.m[1], ref.m[1])
    np.testing.assert_array_equal(mc.m[2], ref.m[2])


@pytest.mark.parametrize("nb,split", [(3, 0), (8, 0), (32, 0), (50, 0), (90, 0), (3, SPLIT), (8, SPLIT), (32, SPLIT), (40, SPLIT),
                                      (8, SPLIT | DUO), (32, SPLIT | DUO)])
def test_flags_order_every_conflicting_pair_and_point_backwards(nb, split):   # (50, 90: bursts of 8 and 16 k tiles)
    tasks, chain, ld, nu = plan(nb, ld=max(nb * T, 4096) if nb == 32 else None, flags=split)
    nt = len(tasks)
    # nodes: bulk tasks 0..nt-1, chain nodes nt + s (s = 2 j: diagonal step, 2 j + 1: sub-diagonal step)
    def chain_node_of_flag(f):  # flag ids >= nt: WD(j) = nt + j -> node nt + 2 j; LSUB(j) = nt + nb + j -> node nt + 2 j + 1
        return nt + 2 * (f - nt) if f < nt + nb else nt + 2 * (f - nt - nb) + 1
    preds = [[] for _ in range(nt + 2 * nb)]
    for i, t in enumerate(tasks):
        assert t.set == i
        # (an upper half's flag goes up after its sibling's: whoever waits for it is ordered behind both halves)
        for d in t.deps + ([t.sibling] if t.sibling != NONE else []):
            if d == NONE:
                continue
            assert d < nt + 2 * nb
            if d < nt:
                preds[i].append(d)
            else:
                preds[i].append(chain_node_of_flag(d))
    for s in range(2 * nb):
        if s > 0:
            preds[nt + s].append(nt + s - 1)
        for c in [chain[s]] + ([chain[2 * nb + s // 2]] if s % 2 == 1 else []):   # (+ the sub-diagonal step's second flag)
            if c != NONE:
                assert c < nt
                preds[nt + s].append(c)
    # the dispatch order: a permutation in which every task dependency points backwards
    order = ORDER[(nb, ld)]
    assert sorted(order) == list(range(nt))
    where = {t: i for i, t in enumerate(order)}
    for i, t in enumerate(tasks):
        for d in t.deps + [t.sibling]:   # (the sibling too: the upper half polls its flag, so somebody must have drawn it)
            if d != NONE and d < nt:
                assert where[d] < where[i], f"task {i} waits for task {d}, which is dispatched AFTER it: deadlock"
    # the dependency graph is acyclic: Kahn's algorithm places every node
    order_preds = [list(p) for p in preds]
    indeg = [len(set(p)) for p in order_preds]
    users = [[] for _ in range(nt + 2 * nb)]
    for n, ps in enumerate(order_preds):
        for q in set(ps):
            users[q].append(n)
    stack = [n for n in range(nt + 2 * nb) if indeg[n] == 0]
    order = []
    while stack:
        n = stack.pop()
        order.append(n)
        for u in users[n]:
            indeg[u] -= 1
            if indeg[u] == 0:
                stack.append(u)
    assert len(order) == nt + 2 * nb, "the dependencies form a cycle"
    # the urgent class: the last burst of every tile and the single-tile products T / E
    for i, t in enumerate(tasks):
        if t.a_mat != 1 or (t.a_mat == 1 and t.b_mat == 2 and False):
            assert i < nu, f"task {i} (T / E) is not on the urgent list"
    # ancestors as bit sets
    anc = [0] * (nt + 2 * nb)
    for n in order:
        a = 0
        for p in preds[n]:
            a |= anc[p] | (1 << p)
        anc[n] = a
    # per tile: readers and writers
    acc = {}
    ch = chain_accesses(nb)
    for n in range(nt + 2 * nb):
        r, w = accesses(tasks[n], ld) if n < nt else ch[n - nt]
        for tile in r:
            acc.setdefault(tile, []).append((n, False))
        for tile in w:
            acc.setdefault(tile, []).append((n, True))
    for tile, lst in acc.items():
        writers = [n for n, wr in lst if wr]
        for wn in writers:
            for n, _ in lst:
                if n == wn:
                    continue
                assert (anc[wn] >> n) & 1 or (anc[n] >> wn) & 1, f"tile {tile}: nodes {wn} and {n} are not ordered"
    # coverage: every product of the left-looking factorisation and of the inverse exactly once
    seen = {}
    for t in tasks:
        oi, oj = tile_of(t.o_off, ld)
        ai, ak = tile_of(t.a_off, ld)
        if t.a_mat == 1 and t.b_mat == 1:      # G (a half-tile task is half of its product)
            for kt in range(t.nk):
                seen[("G", oi, oj, ak + kt)] = seen.get(("G", oi, oj, ak + kt), 0) + (0.5 if t.flags & HALF else 1)
        elif t.a_mat == 1 and t.b_mat == 2:    # X
            for kt in range(t.nk):
                seen[("X", oi, oj, ak + kt)] = seen.get(("X", oi, oj, ak + kt), 0) + 1
    for j in range(nb):
        for i in range(j, nb):
            for k in range(j - 1 if i == j else j):
                assert seen.get(("G", i, j, k)) == 1
    for i in range(1, nb):
        for c in range(i):
            for k in range(c, i):
                assert seen.get(("X", i, c, k)) == 1
    assert sum(seen.values()) == len(seen)


def plan_batch(nb, B):
    """The factor-only plan and the dispatch list of a batched launch of B members (tgp_dag_plan flags: bit 0 + B << 8)."""
    lib = _lib.load()
    ld = nb * T
    flags = 1 | (B << 8)
    n, nu = C.c_int64(), C.c_int64()
    assert lib.tgp_dag_plan(nb, ld, None, 0, C.byref(n), C.byref(nu), None, None, flags) == _lib.TGP_ERR_SHAPE
    tasks = (Task * max(n.value, 1))()
    chain = (C.c_uint32 * (3 * nb))()   # [2 j] diagonal step, [2 j + 1] sub-diagonal step, [2 nb + j] its second flag (split plan)
    order = (C.c_uint32 * max(B * n.value, 1))()
    assert lib.tgp_dag_plan(nb, ld, tasks, n.value, C.byref(n), C.byref(nu), chain, order, flags) == _lib.TGP_OK
    return [tasks[i] for i in range(n.value)], list(chain), ld, nu.value, [order[i] for i in range(B * n.value)]


@pytest.mark.parametrize("nb,B", [(2, 2), (5, 3), (9, 8)])
def test_batched_plan_factors_every_member_from_one_list(nb, B):
    """tgp_nlml_trial_batch: B matrices, B chains, ONE dispatch list of (member << 24 | task) entries drawn in order.
    Executed with a single worker (the worst residency) every member's factor comes out; every member's tasks appear
    exactly once and in an order that is topological for ITS graph (so the in-order dispatcher cannot deadlock whatever
    the other members do); the members' plans are the factor-only plan itself (same tasks, same arithmetic: the values
    equal tgp_nlml_trial's bit for bit)."""
    tasks, chain, ld, nu, order = plan_batch(nb, B)
    single, _, _, _ = plan(nb, flags=1)
    def arithmetic(t):  # what a task computes (not where it sits in the array, nor its flag ids)
        return (t.a_off, t.b_off, t.c_off, t.o_off, t.nk, t.flags, t.a_mat, t.b_mat, t.c_mat, t.o_mat)

    assert sorted(map(arithmetic, tasks)) == sorted(map(arithmetic, single))   # the same tile products, burst for burst
    nt = len(tasks)
    assert len(order) == B * nt
    seen = [[False] * nt for _ in range(B)]
    for e in order:
        b, i = e >> 24, e & 0xFFFFFF
        assert b < B and i < nt and not seen[b][i]
        for dep in tasks[i].deps:
            assert dep == NONE or dep >= nt or seen[b][dep], "a member's entry precedes one of its producers"
        seen[b][i] = True
    assert all(all(s) for s in seen)
    n = nb * T
    mats = [spd(n, 100 + 7 * b + nb) * (1.0 + 0.25 * b) for b in range(B)]
    ORDER[(nb, ld)] = list(range(nt))  # (the members' machines are driven by the merged list below)
    ms = [Machine(A, nb, tasks, chain, ld, nu) for A in mats]
    pos = 0
    while pos < len(order) or any(m.chain_pos < 2 * nb for m in ms):
        progressed = False
        for m in ms:                                  # every member's chain runs whenever it can
            while m.chain_ready():
                m.run_chain()
                progressed = True
        if pos < len(order):
            b, i = order[pos] >> 24, order[pos] & 0xFFFFFF
            if ms[b].ready(tasks[i]):                 # the one worker waits at the head of the list
                ms[b].run_bulk(i)
                ms[b].taken[i] = True
                pos += 1
                progressed = True
        assert progressed, f"stuck at list position {pos} of {len(order)}"
    for A, m in zip(mats, ms):
        L = np.tril(m.m[1])
        np.testing.assert_allclose(L @ L.T, A, rtol=1e-11, atol=1e-11)
    # a member alone (B = 1 machine on the single plan) gives the same bits: members do not interact
    ORDER[(nb, ld)] = ORDER_SINGLE = [e & 0xFFFFFF for e in order if e >> 24 == 0]
    alone = Machine(mats[0], nb, tasks, chain, ld, nu)
    while not alone.done():
        if alone.chain_ready():
            alone.run_chain()
        else:
            alone.run_bulk(alone.acquire())
    np.testing.assert_array_equal(alone.m[1], ms[0].m[1])


@pytest.mark.parametrize("nb,B", [(32, 8), (34, 2), (64, 4)])
def test_batched_dispatch_list_is_topological_at_the_sizes_that_run(nb, B):
    """N = 4096 (eight members per launch), 4352 and 8192: every member's tasks exactly once, every producer of an entry
    earlier in the list or a chain step (the property the deadlock-freedom of the in-order dispatcher rests on), and the
    members interleaved -- no member's work is queued behind another's."""
    tasks, chain, ld, nu, order = plan_batch(nb, B)
    nt = len(tasks)
    assert len(order) == B * nt
    seen = np.zeros((B, nt), dtype=bool)
    first, last = [None] * B, [None] * B
    for pos, e in enumerate(order):
        b, i = e >> 24, e & 0xFFFFFF
        assert b < B and i < nt and not seen[b, i]
        for dep in tasks[i].deps:
            assert dep == NONE or dep >= nt or seen[b, dep]
        seen[b, i] = True
        first[b] = pos if first[b] is None else first[b]
        last[b] = pos
    assert seen.all()
    assert max(first) < B * nt // 4 and min(last) > B * nt // 2     # all members are in flight together
