"""Development aid: summarise the time stamps of one persistent `update` (TGP_DAG_TRACE=<file> python tools/dag_trace.py N).
Chain: per step the wait before the leaf, the diagonal product, the leaf (+ W store), the wait before L(j+1,j), the
sub-diagonal product.  Tasks: wait for flags / run time per kind, workgroup utilisation over the launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def duo_summary(N, NB, nt, ch, tk):
    """The two-workgroup chain (round 6): per step j the helper's stamps -- [0] entry, [1] P(j,j-1) / P(j,j) flags up, [8 + kc] column
    block kc started, [16 + kc] finished -- and the leaf's: [2] start, [3] end (W_jj published)."""
    t0 = ch[0, 2]
    end = max(ch[-1, 3], tk[:, 2].max())
    print(f"N={N} NB={NB} tasks={nt} (two-workgroup chain): launch span {end - t0:.1f} us; last leaf ends at {ch[-1, 3] - t0:.1f} us")
    step = np.diff(ch[:, 2])
    leaf = ch[:, 3] - ch[:, 2]
    wait_bulk = ch[1:, 1] - ch[1:, 0]
    gap = ch[1:, 2] - ch[:-1, 3]                      # leaf(j) starts this long after leaf(j-1) ended
    idle = ch[2:, 0] - ch[:-2, 3]                     # the workgroup's own turn-round: leaf(j-2) end -> helper(j) entry
    first = ch[1:, 8] - ch[1:, 1]                     # operands + panel 0 staged
    work = ch[1:, 16:24] - ch[1:, 8:16]               # per column block: compute (incl. waiting for the next panel's flag)
    between = ch[1:, 9:16] - ch[1:, 16:23]            # end barrier
    print(f"step {step.mean():.1f} us (min {step.min():.1f} max {step.max():.1f});  leaf {leaf.mean():.1f};  leaf(j) starts {gap.mean():.1f} us after leaf(j-1) ends "
          f"(min {gap.min():.1f} max {gap.max():.1f})")
    print(f"helper: waits for the bulk's tiles {wait_bulk.mean():.1f} us (max {wait_bulk.max():.1f}); loads + panel 0 staged {first.mean():.1f}; "
          f"column blocks {np.round(work.mean(0), 1)} (sum {work.sum(1).mean():.1f}); barriers between {between.mean():.2f}")
    late = ch[1:, 1] - ch[:-1, 2]                    # when the helper has its tiles, relative to the start of the leaf it follows
    print(f"  the helper has P(j,j-1) {late.mean():.1f} us after leaf(j-1) started (min {late.min():.1f} max {late.max():.1f}); "
          f"its last column block ends {(ch[1:, 23] - ch[:-1, 3]).mean():.1f} us after that leaf ended")
    print("  first steps (step, wait-bulk, gap):", " | ".join(f"{a:.0f} {b:.0f} {c:.0f}" for a, b, c in zip(step[:8], wait_bulk[:8], gap[:8])))
    # the tasks the helper waits for (plan flags 2 | 4: split + two-workgroup chain; chain_dep is [5 NB] then)
    import ctypes as C
    from trieste_amd import _lib
    lib = _lib.load()

    class Task(C.Structure):
        _fields_ = [("a_off", C.c_uint32), ("b_off", C.c_uint32), ("c_off", C.c_uint32), ("o_off", C.c_uint32),
                    ("nk", C.c_uint32), ("flags", C.c_uint32), ("a_mat", C.c_uint8), ("b_mat", C.c_uint8),
                    ("c_mat", C.c_uint8), ("o_mat", C.c_uint8), ("dep", C.c_uint32 * 3), ("set", C.c_uint32),
                    ("dep3", C.c_uint32)]
    n_, nu_ = C.c_int64(), C.c_int64()
    lib.tgp_dag_plan(NB, NB * 128, None, 0, C.byref(n_), C.byref(nu_), None, None, 6)
    tarr = (Task * n_.value)()
    carr = (C.c_uint32 * (3 * NB))()
    lib.tgp_dag_plan(NB, NB * 128, tarr, n_.value, C.byref(n_), C.byref(nu_), carr, None, 6)
    tid = {(t.o_off, t.flags & 16): i for i, t in enumerate(tarr) if t.a_mat == 0}   # T tasks by output tile (and half)
    for j in (6, 7, 14, 15, 24, 25):
        z = ch[j - 1, 2]   # leaf(j-1) starts
        line = [f"  step {j}: relative to the start of leaf({j - 1}): helper({j - 1}) block 5 at {ch[j - 1, 13] - z:.1f}, its end {ch[j - 1, 23] - z:.1f};"]
        ld = NB * 128
        o_t = j * 128 * ld + (j - 2) * 128
        for name, t in (("G(j,j-1) lo", carr[2 * j - 1]), ("hi", carr[2 * NB + j - 1]), ("T(j,j-2) lo", tid.get((o_t, 0), 0xFFFFFFFF)),
                        ("hi", tid.get((o_t, 16), 0xFFFFFFFF)), ("G(j,j)", carr[2 * j])):
            if t == 0xFFFFFFFF: continue
            line.append(f"{name}: drawn {tk[t, 0] - z:.1f} started {tk[t, 1] - z:.1f} ended {tk[t, 2] - z:.1f};")
        g = carr[2 * j - 1]
        if g != 0xFFFFFFFF:
            for d in list(tarr[g].dep) + [tarr[g].dep3]:
                if d == 0xFFFFFFFF: continue
                if d < nt: line.append(f"[dep task {d}: out=({tarr[d].o_off // (ld * 128)},{(tarr[d].o_off % ld) // 128}) nk={tarr[d].nk} drawn {tk[d, 0] - z:.1f} started {tk[d, 1] - z:.1f} ended {tk[d, 2] - z:.1f}]")
                else: line.append(f"[dep chain flag {d - nt}]")
        line.append(f"helper({j}) has its tiles at {ch[j, 1] - z:.1f}, first block at {ch[j, 8] - z:.1f}, done {ch[j, 23] - z:.1f}; leaf({j - 1}) ends {ch[j - 1, 3] - z:.1f}")
        print(" ".join(line))
    wait, run = tk[:, 1] - tk[:, 0], tk[:, 2] - tk[:, 1]
    nwg = len(set(tk[:, 3].astype(int)))
    print(f"bulk: {nwg} workgroups ran tasks; busy {run.sum():.0f} us = {run.sum() / (nwg * (end - t0)):.2f} of (workgroups x span); "
          f"waiting on flags {wait.sum():.0f} us")


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    path = os.environ.get("TGP_DAG_TRACE", "/tmp/dag_trace.bin")
    os.environ["TGP_DAG_TRACE"] = path
    from trieste_amd import objectives as O
    from trieste_amd.engine import GPEngine
    import ctypes as C
    from trieste_amd import _lib
    X, Y = O.synthetic_problem(O.ackley, 8, N)
    eng = GPEngine(8, "matern52")
    eng.set_variant(32 | int(os.environ.get('TGP_VARIANT', '0')))   # (TGP_VARIANT=256: the plan without round 6's split)
    eng.set_hyper(1.0, O.default_lengthscales(8), 1e-2, float(Y.mean()))
    for _ in range(3):
        eng.set_data(X, Y)
    raw = np.fromfile(path, dtype=np.uint64)
    NB, nt = int(raw[0]), int(raw[1])
    ch = raw[2:2 + 32 * NB].reshape(NB, 32).astype(np.float64) * 1e-2      # us (100 MHz clock)
    tk = raw[2 + 32 * NB:].reshape(nt, 4).astype(np.float64)
    tk[:, :3] *= 1e-2
    variant = int(os.environ.get('TGP_VARIANT', '0'))
    split = 3 <= NB < 48 and not (variant & 256)
    duo = split and not (variant & 512)
    if duo:
        return duo_summary(N, NB, nt, ch, tk)
    t0 = ch[0, 0]
    end = max(ch[-1, 3], tk[:, 2].max())
    print(f"N={N} NB={NB} tasks={nt}: launch span {end - t0:.1f} us; chain ends at {ch[-1, 3] - t0:.1f} us")
    waitA, diag, leaf = ch[:, 1] - ch[:, 0], ch[:, 2] - ch[:, 1], ch[:, 3] - ch[:, 2]
    waitB, sub = ch[:-1, 4] - ch[:-1, 3], ch[:-1, 5] - ch[:-1, 4]
    print(f"chain per step (us): wait-diag {waitA.mean():.1f} (max {waitA.max():.1f})  diag {diag[1:].mean():.1f}  leaf {leaf.mean():.1f}  "
          f"wait-sub {waitB.mean():.1f} (max {waitB.max():.1f})  sub {sub.mean():.1f};  step {np.diff(ch[:, 0]).mean():.1f}")
    dm, sm = ch[1:, 6] - ch[1:, 1], ch[:-1, 7] - ch[:-1, 4]
    print(f"  inside: diag entry..end of MFMA loop {dm.mean():.1f}, rest {(diag[1:] - dm).mean():.1f};  sub entry..end of MFMA {sm.mean():.1f}, rest {(sub - sm).mean():.1f}")
    mid = slice(4, NB - 1)
    print("  sub, per wave, relative to its entry (mean over steps): loads landed", np.round((ch[mid, 16:24] - ch[mid, 4:5]).mean(0), 1),
          " MFMAs done", np.round((ch[mid, 8:16] - ch[mid, 4:5]).mean(0), 1))
    print("  diag, per wave, MFMA loop done after entry:", np.round((ch[mid, 24:32] - ch[mid, 1:2]).mean(0), 1))
    print("  first steps:", " | ".join(f"{a:.0f} {b:.0f} {c:.0f} {d:.0f} {e:.0f}" for a, b, c, d, e in
                                      zip(waitA[:6], diag[:6], leaf[:6], waitB[:6], sub[:6])))
    # task kinds from the plan
    lib = _lib.load()

    class Task(C.Structure):
        _fields_ = [("a_off", C.c_uint32), ("b_off", C.c_uint32), ("c_off", C.c_uint32), ("o_off", C.c_uint32),
                    ("nk", C.c_uint32), ("flags", C.c_uint32), ("a_mat", C.c_uint8), ("b_mat", C.c_uint8),
                    ("c_mat", C.c_uint8), ("o_mat", C.c_uint8), ("dep", C.c_uint32 * 3), ("set", C.c_uint32),
                    ("dep3", C.c_uint32)]

    ld = NB * 128
    # the plan the engine used: the round-6 split plan at the chain-bound sizes unless tgp_set_variant bit 8 switched it off
    PLAN_FLAGS = 2 if split else 0
    n_, nu_ = C.c_int64(), C.c_int64()
    lib.tgp_dag_plan(NB, ld, None, 0, C.byref(n_), C.byref(nu_), None, None, PLAN_FLAGS)
    tarr = (Task * n_.value)()
    carr = (C.c_uint32 * (3 * NB))()
    lib.tgp_dag_plan(NB, ld, tarr, n_.value, C.byref(n_), C.byref(nu_), carr, None, PLAN_FLAGS)
    tasks, chain, nu = [tarr[i] for i in range(n_.value)], list(carr), nu_.value
    kinds = []
    for t in tasks:
        if t.a_mat == 1 and t.b_mat == 1: kinds.append("G%d" % t.nk)
        elif t.a_mat == 0: kinds.append("T")
        elif t.a_mat == 1: kinds.append("X%d" % t.nk)
        else: kinds.append("E")
    kinds = np.array(kinds)
    wait, run = tk[:, 1] - tk[:, 0], tk[:, 2] - tk[:, 1]
    for k in sorted(set(kinds)):
        m = kinds == k
        print(f"  {k:3s} n={m.sum():5d}  run {run[m].mean():6.1f} us (min {run[m].min():.1f})  wait {wait[m].mean():6.1f} us (max {wait[m].max():.0f})")
    # the chain's late inputs: who was late, and why
    for jj in np.argsort(-waitB)[:3]:
        dep = chain[2 * jj + 1]
        if dep == 0xFFFFFFFF: continue
        def line(i, ind="    "):
            t = tasks[i]
            return (f"{ind}task {i} {kinds[i]} out=({t.o_off // (ld * 128)},{(t.o_off % ld) // 128}) popped {tk[i,0]-t0:.0f} started {tk[i,1]-t0:.0f} "
                    f"ended {tk[i,2]-t0:.0f} wg {int(tk[i,3])}")
        print(f"  step {jj}: chain waited {waitB[jj]:.0f} us from {ch[jj,3]-t0:.0f} for")
        print(line(dep))
        for d in tasks[dep].dep:
            if d == 0xFFFFFFFF: continue
            if d < nt:
                print(line(d, "      <- "))
                for d2 in tasks[d].dep:
                    if d2 != 0xFFFFFFFF and d2 < nt: print(line(d2, "          <- "))
                    elif d2 != 0xFFFFFFFF: print(f"          <- chain flag {d2 - nt} (WD/LSUB) at {ch[(d2-nt) % NB, 3 if d2 - nt < NB else 6]-t0:.0f}")
            else:
                print(f"      <- chain flag {d - nt}")
    busy = run.sum()
    nwg = len(set(tk[:, 3].astype(int)))
    print(f"bulk: {nwg} workgroups ran tasks; busy {busy:.0f} us = {busy / (nwg * (end - t0)):.2f} of (workgroups x span); "
          f"waiting on flags {wait.sum():.0f} us")

main()
