"""Phase timing of the int8 sweep's step (development aid).  Needs a build with -DTGP_I8_TRACE=1 (tools/build_exp.sh):
   TGP_LIB=tools/exp/libtgp_i8tr.so python tools/i8_trace.py [i8x4|i8x5]
Stamps (shader cycles, per wave): 0 step start, 1 DMA issued, 2 first fragment's MFMAs issued, 3 all MFMAs issued,
4 generation / row-block fold done, 5 DMA wait done, 6 barrier passed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trieste_amd import objectives as O, _lib
from trieste_amd.engine import GPEngine

prec = sys.argv[1] if len(sys.argv) > 1 else "i8x4"
N, d, M = 4096, 8, 1 << 17
X, Y = O.synthetic_problem(O.ackley, d, N)
eng = GPEngine(d, "matern52")
eng.set_hyper(1.0, O.default_lengthscales(d), 1e-2, float(Y.mean()))
eng.set_data(X, Y)
eta = eng.eta()
Xq = eng.sample_box(5678, 0, M, 0.0, 1.0)
eng.set_precision(prec)
for _ in range(2):
    eng.acq_argmax("ei", eta, Xq)
print("kernel ms", eng.last_kernel_ms()[0], "for", M, "candidates")
lib = ctypes.CDLL(_lib.LIB_PATH)
NT, S = 128, 12
buf = np.zeros((2, NT, 8, S), dtype=np.uint64)
rc = lib.tgp_dev_i8_trace(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = ["dma issue", "frag0 mfma", "frag1 mfma", "gen/fold", "vm wait", "barrier"]
for wg in range(2):
    t = buf[wg].astype(np.int64)
    info = t[:, :, 7]
    gen = (info >> 32) & 1
    diag = (info >> 16) & 1
    st = t[:, :, :7]
    st_all = t
    step = st[:, :, 6] - st[:, :, 0]
    nxt = st[1:, :, 0] - st[:-1, :, 6]
    print(f"== workgroup {wg}: step (stamp 0 -> 6) mean {step.mean():.0f} cycles, median {np.median(step):.0f}; "
          f"loop overhead between steps {nxt.mean():.0f}; full period {(st[-1, 0, 0] - st[0, 0, 0]) / (NT - 1):.0f}")
    for label, mask in (("plain steps", (gen[:, 0] == 0) & (diag[:, 0] == 0)), ("generating steps (tile t+1 generated)", gen[:, 0] == 1),
                        ("diagonal-block steps", diag[:, 0] == 1)):
        if not mask.any():
            continue
        sel = st[mask]
        print(f"  {label}: {mask.sum()} steps, mean step {(sel[:, :, 6] - sel[:, :, 0]).mean():.0f}")
        for i, nm in enumerate(names):
            a, b = sel[:, :, i], sel[:, :, i + 1]
            if i == 1:  # frag 0 may be skipped (stamp 0)
                ok = b > 0
                dd = np.where(ok, b - a, 0)
            elif i == 2:
                a = np.where(a > 0, a, sel[:, :, 1])
                dd = b - a
            else:
                dd = b - a
            print(f"     {nm:12s} mean {dd.mean():7.0f}   per wave " + " ".join(f"{x:6.0f}" for x in dd.mean(axis=0)))
    gsel = st_all[gen[:, 0] == 1]
    if gsel.size:
        g0, g1, g2 = gsel[:, :, 8], gsel[:, :, 9], gsel[:, :, 10]
        ok = g0 > 0
        print("  inside generate_B, per wave: start after step start", " ".join(f"{x:6.0f}" for x in np.where(ok, g0 - gsel[:, :, 0], 0).mean(axis=0)))
        print("     distances (scalar loads, LDS reads, 16 x d FMAs)  ", " ".join(f"{x:6.0f}" for x in np.where(ok, g1 - g0, 0).mean(axis=0)))
        print("     4 x kernel + digits + LDS writes                 ", " ".join(f"{x:6.0f}" for x in np.where(ok, g2 - g1, 0).mean(axis=0)))
    # skew: when does each wave arrive at the barrier relative to the last
    arr = st[:, :, 5] - st[:, :, 5].max(axis=1, keepdims=True)
    print("  arrival at the barrier relative to the last wave, per wave:", " ".join(f"{x:6.0f}" for x in arr.mean(axis=0)))
    rel = st[:, :, 6] - st[:, :, 6].min(axis=1, keepdims=True)
    print("  release skew per wave:", " ".join(f"{x:6.0f}" for x in rel.mean(axis=0)))
