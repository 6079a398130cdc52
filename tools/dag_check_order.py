"""Development aid: run updates with TGP_DAG_TRACE until one FAILS, then check the recorded time stamps: did every
task start after its producers ended, did the chain read its inputs after their producers ended?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
path = "/tmp/dag_trace.bin"
os.environ["TGP_DAG_TRACE"] = path
os.environ["TGP_DAG_DUMP"] = "/tmp/dag_A.bin"
from trieste_amd import objectives as O, _lib
from trieste_amd.engine import GPEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = 4
X, Y = O.synthetic_problem(O.ackley, d, N)
ls = O.default_lengthscales(d)
old = GPEngine(d, "matern52"); old.set_variant(16); old.set_hyper(1.0, ls, 1e-2, float(Y.mean())); old.set_data(X, Y)
L0, _, _ = old.get_factor()
lib = _lib.load()
class Task(C.Structure):
    _fields_ = [("a_off", C.c_uint32), ("b_off", C.c_uint32), ("c_off", C.c_uint32), ("o_off", C.c_uint32),
                ("nk", C.c_uint32), ("flags", C.c_uint32), ("a_mat", C.c_uint8), ("b_mat", C.c_uint8),
                ("c_mat", C.c_uint8), ("o_mat", C.c_uint8), ("dep", C.c_uint32 * 3), ("set", C.c_uint32), ("dep3", C.c_uint32)]
for attempt in range(60):
    eng = GPEngine(d, "matern52"); eng.set_variant(32); eng.set_hyper(1.0, ls, 1e-2, float(Y.mean()))
    failed = False
    try:
        eng.set_data(X, Y)
        L, _, _ = eng.get_factor()
        failed = not np.allclose(L, L0, atol=1e-8)
    except Exception as e:
        failed = True
        print("  exception:", str(e)[:200])
    if not failed:
        continue
    raw = np.fromfile(path, dtype=np.uint64)
    NB, nt = int(raw[0]), int(raw[1])
    ch = raw[2:2 + 32 * NB].reshape(NB, 32).astype(np.int64)
    tk = raw[2 + 32 * NB:].reshape(nt, 4).astype(np.int64)
    ld = NB * 128
    # the plan the engine used: the round-6 split plan at the chain-bound sizes unless tgp_set_variant bit 8 switched it off
    PLAN_FLAGS = 2 if (3 <= NB < 48 and not (int(os.environ.get('TGP_VARIANT', '0')) & 256)) else 0
    n_, nu_ = C.c_int64(), C.c_int64()
    lib.tgp_dag_plan(NB, ld, None, 0, C.byref(n_), C.byref(nu_), None, None, PLAN_FLAGS)
    tarr = (Task * n_.value)(); carr = (C.c_uint32 * (3 * NB))()
    lib.tgp_dag_plan(NB, ld, tarr, n_.value, C.byref(n_), C.byref(nu_), carr, None, PLAN_FLAGS)
    t0 = ch[0, 0]
    print(f"attempt {attempt} FAILED; NB={NB} tasks={nt}")
    bad = 0
    def ev_time(f):  # chain flag -> time it was published (approx: stamp after leaf / end of next diag's loop)
        e = f - nt
        return ch[e, 3] if e < NB else ch[e - NB + 1, 0]  # (earliest possible: the slow path publishes before the step's product)
    for i in range(nt):
        for dpd in tarr[i].dep:
            if dpd == 0xFFFFFFFF: continue
            te = tk[dpd, 2] if dpd < nt else ev_time(dpd)
            if tk[i, 1] < te:
                bad += 1
                if bad < 10: print(f"  task {i} started {tk[i,1]-t0} before dep {dpd} ended {te-t0}")
    for j in range(NB):
        for part, slot in ((0, 1), (1, 4)):
            dep = carr[2 * j + part]
            if dep == 0xFFFFFFFF or (part == 1 and j + 1 >= NB): continue
            if ch[j, slot] < tk[dep, 2]:
                bad += 1
                print(f"  chain step {j} part {part} read at {ch[j,slot]-t0} before task {dep} ended {tk[dep,2]-t0}")
    Npad = NB * 128
    both = np.fromfile("/tmp/dag_A.bin", dtype=np.float64).reshape(2, Npad, Npad)
    Ad, Ld = both[0], both[1]
    Lp = np.zeros((Npad, Npad)); Lp[:N, :N] = np.tril(L0); Lp[N:, N:] = np.eye(Npad - N)
    Kp = Lp @ Lp.T
    Tt = 128
    def bk(M, i, j): return M[i*Tt:(i+1)*Tt, j*Tt:(j+1)*Tt]
    wrongP = []
    for j in range(NB):
        for i in range(j, NB):
            hi = j - 1 if i == j else j
            exp = bk(Kp, i, j) - sum(bk(Lp, i, k) @ bk(Lp, j, k).T for k in range(max(hi, 0)))
            got = bk(Ad, i, j)
            err = np.abs(np.tril(got - exp) if i == j else got - exp)
            err = np.where(np.isnan(err), np.inf, err)
            if err.max() > 1e-8:
                wrongP.append((i, j, float(err.max())))
                if len(wrongP) <= 2:
                    bad_r, bad_c = np.where(err > 1e-8)
                    print(f"  P({i},{j}) WRONG: max err {err.max():.3g}; {len(bad_r)} entries; rows {bad_r.min()}..{bad_r.max()} cols {bad_c.min()}..{bad_c.max()}")
                    for k in range(max(hi, 0)):
                        cc = bk(Lp, i, k) @ bk(Lp, j, k).T
                        e2 = np.abs(np.tril(got - exp - cc) if i == j else got - exp - cc).max()
                        e3 = np.abs(np.tril(got - exp + cc) if i == j else got - exp + cc).max()
                        if min(e2, e3) < 1e-8: print(f"    = expected {'+' if e2 < e3 else '-'} contribution of column {k}")
    wl = []
    for j in range(NB):
        for i in range(j, NB):
            e = np.abs(np.tril(bk(Ld, i, j) - bk(Lp, i, j)) if i == j else bk(Ld, i, j) - bk(Lp, i, j))
            e = np.where(np.isnan(e), np.inf, e)
            if e.max() > 1e-8: wl.append((i, j, round(float(e.max()), 4)))
    print("  wrong L tiles in memory (i, j, err), column-major:", wl[:10])
    if wl and wl[0][0] == wl[0][1]:
        j = wl[0][0]
        Lj = np.tril(bk(Ld, j, j))
        Seff = Lj @ Lj.T
        Strue = bk(Kp, j, j) - sum(bk(Lp, j, k) @ bk(Lp, j, k).T for k in range(j))
        print(f"  leaf input S({j},{j}) reconstructed from the stored factor vs truth, max |diff| per 16 x 16 block (log10, '.' < 1e-9):")
        for bi in range(8):
            row = []
            for bj in range(bi + 1):
                e = np.abs(Seff[16*bi:16*bi+16, 16*bj:16*bj+16] - Strue[16*bi:16*bi+16, 16*bj:16*bj+16])
                e = np.where(np.isnan(e), np.inf, e).max()
                row.append("   ." if e < 1e-9 else (" inf" if not np.isfinite(e) else f"{np.log10(e):4.0f}"))
            print("   ", " ".join(row))
        # is the wrong S the bulk partial sum WITHOUT the chain's own column, or with a stale one?
        Pmem = bk(Ad, j, j)
        if j >= 1:
            for name, cand in (("P(j,j) as stored (column j-1 not subtracted)", Pmem),
                               ("P(j,j) - 2 Lsub Lsub^T", Pmem - 2 * bk(Lp, j, j - 1) @ bk(Lp, j, j - 1).T)):
                print(f"    |S_eff - {name}| = {np.abs(np.tril(Seff - cand)).max():.3g}")
            if j >= 2:
                cand = Pmem - bk(Lp, j - 1, j - 2) @ bk(Lp, j - 1, j - 2).T
                print(f"    |S_eff - (P(j,j) - L(j-1,j-2) L(j-1,j-2)^T)| (the PREVIOUS step's Lsub) = {np.abs(np.tril(Seff - cand)).max():.3g}")
    print("  wrong P tiles (i, j, err), column-major:", [(i, j, round(e, 4)) for i, j, e in wrongP[:10]])
    print("  chain waits per step (waitA, waitB) in 10 ns:", [(int(ch[j,1]-ch[j,0]), int(ch[j,4]-ch[j,3])) for j in range(NB)])
    never = [i for i in range(nt) if tk[i, 2] == 0]
    print(f"  ordering violations: {bad}; tasks never run: {len(never)} {never[:10]}")
    # which tile is wrong first
    T = 128
    wrong = [(i, j) for i in range(NB) for j in range(i + 1) if not np.allclose(L[i*T:(i+1)*T, j*T:(j+1)*T], L0[i*T:(i+1)*T, j*T:(j+1)*T], atol=1e-8)] if 'L' in dir() else []
    print("  wrong L tiles:", wrong[:6])
    break
else:
    print("no failure in 60 attempts")
