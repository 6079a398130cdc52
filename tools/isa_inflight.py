"""Development aid / test helper: the W-operand register sets of sweep_i8a_kernel are loaded by inline-asm global_load_dwordx4
and are in flight for a step or two -- the compiler, which believes the values are there when the asm returns, must never copy,
spill or reuse those registers.  For every sweep_i8a_kernel in a `hipcc -S` file: the registers that the asm loads inside the
step loop target (= loaded at more than one site) may be named only by those loads, by MFMAs as their A operand, and as the
SOURCE of the v_mov_b32 of the three-set form's copy.   usage: python tools/isa_inflight.py <file.s>   (exit 1 on a violation)"""
import re
import sys
from collections import Counter


def _refs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check(path):
    """-> list of (kernel, number of in-flight registers, [(line, text) violations])"""
    txt = open(path).read()
    res = []
    for f in re.split(r"\n(?=_ZN3tgp[^\n]*: )", txt):
        name = f.split(":", 1)[0]
        if "sweep_i8a_kernel" not in name:
            continue
        body = f.split("s_endpgm")[0].split("\n")
        sites = Counter()
        for l in body:
            m = re.search(r"^\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[", l)
            if m:
                for r in range(int(m.group(1)), int(m.group(2)) + 1):
                    sites[r] += 1
        regs = {r for r, n in sites.items() if n > 1}
        bad = []
        for i, l in enumerate(body):
            t = l.strip()
            if not t or t[0] in ";.":
                continue
            if not (_refs(t) & regs):
                continue
            op, _, rest = t.partition(" ")
            ops = [x.strip() for x in rest.split(",")]
            if op == "global_load_dwordx4" and not (_refs(",".join(ops[1:])) & regs):
                continue
            if op == "v_mfma_i32_32x32x32_i8" and not ((_refs(ops[0]) | _refs(ops[2]) | _refs(ops[3])) & regs):
                continue
            if op == "v_mov_b32" and not (_refs(ops[0]) & regs):
                continue
            bad.append((i + 1, t))
        res.append((name, len(regs), bad))
    return res


if __name__ == "__main__":
    rc = 0
    for name, n, bad in check(sys.argv[1]):
        print(f"{name[-44:]}: {n} in-flight registers, {len(bad)} violations")
        for b in bad[:10]:
            print("    ", b)
        rc |= bool(bad) or n not in (32,)
    sys.exit(rc)
