"""Development aid (round 6): execute the REAL plan (tgp_dag_plan) of the persistent `update` kernel on a host model of the launch --
workers drawing the list in order and waiting for the flags of what they drew, the chain as one workgroup or as two (leaf / helper)
-- with the durations measured on the device (profiles/r06_dag_duo_v8.txt), to see what a change of the plan's ORDER does to the
chain's step before any GPU time is spent.   usage: python tools/dag_exec_sim.py [NB=32] [plan flags=6] [workers=254]"""
import ctypes as C, heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import _lib

NONE = 0xFFFFFFFF
FLAG_LAT = 1.5          # a flag is seen this long after it was set
LEAF, HELP_LOAD, HELP_BLOCKS, HELP_TAIL = 27.3, 6.5, 24.0, 2.0
ONE = dict(diag=13.5, leaf=26.2, sub=14.2)


class Task(C.Structure):
    _fields_ = [("a_off", C.c_uint32), ("b_off", C.c_uint32), ("c_off", C.c_uint32), ("o_off", C.c_uint32),
                ("nk", C.c_uint32), ("flags", C.c_uint32), ("a_mat", C.c_uint8), ("b_mat", C.c_uint8),
                ("c_mat", C.c_uint8), ("o_mat", C.c_uint8), ("dep", C.c_uint32 * 3), ("set", C.c_uint32), ("dep3", C.c_uint32)]


def plan(NB, flags):
    lib = _lib.load()
    n_, nu_ = C.c_int64(), C.c_int64()
    lib.tgp_dag_plan(NB, C.c_int64(NB * 128), None, C.c_int64(0), C.byref(n_), C.byref(nu_), None, None, flags)
    tarr = (Task * n_.value)(); carr = (C.c_uint32 * (3 * NB))(); order = (C.c_uint32 * (n_.value + 1))()
    assert lib.tgp_dag_plan(NB, C.c_int64(NB * 128), tarr, C.c_int64(n_.value), C.byref(n_), C.byref(nu_), carr, order, flags) == 0
    return [tarr[i] for i in range(n_.value)], list(carr), [order[i] for i in range(n_.value)]


def dur(t):
    if t.flags & 8: return 13.8                       # half-tile task
    if t.a_mat == 1 and t.b_mat == 2: return 6.4 + 17.0 * t.nk   # X
    return 5.3 + 18.3 * t.nk


def run(NB, flags, workers, verbose=True):
    tasks, chain, order = plan(NB, flags)
    nt = len(tasks)
    duo = bool(flags & 4)
    WD, LSUB = nt, nt + NB
    t_set = np.full(nt + 2 * NB, np.inf)               # when each flag is set
    # The chain's flags depend on bulk flags and vice versa: iterate the whole launch as a discrete-event simulation with the chain
    # advanced lazily (its next event is computed whenever the flags it waits for are known).
    free = [(0.0, w) for w in range(workers)]
    heapq.heapify(free)
    head = 0
    drawn = np.zeros(nt); started = np.zeros(nt); ended = np.zeros(nt)
    leaf_start = np.full(NB, np.inf); leaf_end = np.full(NB, np.inf); help_in = np.full(NB, np.inf)

    def seen(f):
        return 0.0 if f == NONE else t_set[f] + FLAG_LAT

    def advance_chain():
        """set every chain flag whose inputs are known"""
        changed = True
        while changed:
            changed = False
            for j in range(NB):
                if np.isfinite(leaf_end[j]): continue
                if duo:
                    if j == 0:
                        s = 3.0
                    else:
                        if not np.isfinite(leaf_start[j - 1]): break
                        need = [chain[2 * j - 1], chain[2 * NB + j - 1]]
                        if any(f != NONE and not np.isfinite(t_set[f]) for f in need): break
                        p_in = max([seen(f) for f in need] + [leaf_end[j - 2] if j >= 2 else 0.0])
                        d = chain[2 * j]
                        if d != NONE and not np.isfinite(t_set[d]): break
                        help_in[j] = p_in
                        s = max(p_in + HELP_LOAD + HELP_BLOCKS, leaf_end[j - 1] + 3.0, seen(d)) + HELP_TAIL
                        t_set[LSUB + j - 1] = s - HELP_TAIL
                    leaf_start[j] = s
                    leaf_end[j] = s + LEAF
                    t_set[WD + j] = leaf_end[j]
                    changed = True
                else:
                    d = chain[2 * j]
                    if d != NONE and not np.isfinite(t_set[d]): break
                    prev = 0.0 if j == 0 else sub_end[j - 1]
                    if j > 0 and not np.isfinite(prev): break
                    s = max(prev, seen(d)) + (3.0 if j == 0 else ONE["diag"])
                    leaf_start[j] = s
                    leaf_end[j] = s + ONE["leaf"]
                    t_set[WD + j] = leaf_end[j]
                    if j + 1 < NB:
                        need = [chain[2 * j + 1], chain[2 * NB + j]]
                        if any(f != NONE and not np.isfinite(t_set[f]) for f in need):
                            pending_sub.append(j)
                        else:
                            sub_end[j] = max([leaf_end[j]] + [seen(f) for f in need]) + ONE["sub"]
                            t_set[LSUB + j] = sub_end[j] + ONE["diag"]   # (published under the next diagonal product)
                    changed = True
            for j in list(pending_sub):
                need = [chain[2 * j + 1], chain[2 * NB + j]]
                if all(f == NONE or np.isfinite(t_set[f]) for f in need):
                    sub_end[j] = max([leaf_end[j]] + [seen(f) for f in need]) + ONE["sub"]
                    t_set[LSUB + j] = sub_end[j] + ONE["diag"]
                    pending_sub.remove(j)
                    changed = True

    sub_end = np.full(NB, np.inf); pending_sub = []
    advance_chain()
    # workers: each pops the next list entry when free, waits for its flags (known or not yet: resolve lazily)
    waiting = []   # (task, worker, drawn time) whose deps are not all known yet
    now = 0.0
    while head < nt or waiting:
        progressed = False
        # resolve waiting tasks whose deps are all known
        for item in list(waiting):
            i, w, td = item
            t = tasks[i]
            sib = bool(t.flags & 32)   # DAG_SIB: dep3 is the lower-half sibling, waited for at the END
            deps = [d for d in list(t.dep) + ([] if sib else [t.dep3]) if d != NONE]
            if all(np.isfinite(t_set[d]) for d in deps) and (not sib or np.isfinite(t_set[t.dep3])):
                st = max([td] + [seen(d) for d in deps])
                started[i] = st; ended[i] = st + dur(t); t_set[t.set] = max(ended[i], seen(t.dep3)) if sib else ended[i]
                heapq.heappush(free, (ended[i], w))
                waiting.remove(item)
                progressed = True
        advance_chain()
        if head < nt and free:
            tf, w = heapq.heappop(free)
            i = order[head]; head += 1
            drawn[i] = tf
            waiting.append((i, w, tf))
            progressed = True
        if not progressed:
            raise RuntimeError(f"stuck at head {head}, {len(waiting)} waiting")
    advance_chain()
    step = np.diff(leaf_start)
    if verbose:
        print(f"NB={NB} flags={flags} workers={workers}: launch {max(leaf_end.max(), ended.max()):.0f} us; step {step.mean():.1f} (min {step.min():.1f} max {step.max():.1f})")
        print("  steps:", " ".join(f"{x:.0f}" for x in step))
        if duo:
            late = help_in[1:] - leaf_start[:-1]
            print("  the helper has its tiles this long after the leaf it follows started:", " ".join(f"{x:.0f}" for x in late))
    return dict(step=step, tasks=tasks, drawn=drawn, started=started, ended=ended, leaf_start=leaf_start, help_in=help_in)


def tile_history(NB, flags, workers, j):
    """debug: every task writing tile (j, j-1), (j, j-2) and (j, j) with its times relative to the start of leaf(j-1)"""
    r = run(NB, flags, workers, verbose=False)
    ld = NB * 128
    z = r["leaf_start"][j - 1]
    for i, t in enumerate(r["tasks"]):
        oi, oj = t.o_off // (ld * 128), (t.o_off % ld) // 128
        if (oi, oj) in ((j, j - 1), (j, j), (j, j - 2)) and t.o_mat in (0, 1):
            kind = "T" if t.a_mat == 0 else "G"
            k0 = (t.a_off % ld) // 128 if kind == "G" else -1
            print(f"  {kind}({oi},{oj}) k0={k0} nk={t.nk} half={(t.flags >> 3) & 3}: drawn {r['drawn'][i] - z:.0f} started {r['started'][i] - z:.0f} ended {r['ended'][i] - z:.0f}")


if __name__ == "__main__":
    NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else (254 if flags & 4 else 255)
    run(NB, flags, workers)
