"""Development aid: per-kernel register / scratch statistics from the assembly hipcc leaves with -save-temps, and
where the scratch accesses sit relative to the loops (a scratch reload inside a loop that has DMA or loads in flight is
an s_waitcnt vmcnt(0) on them).   usage: python tools/isa_spills.py <file.s> [name-substring]"""
import re, subprocess, sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
s = open(path).read()
md = s[s.index("amdhsa.kernels:"):]
def filt(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
for b in md.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    name = g("name")
    dn = filt(name)
    if want and want not in dn:
        continue
    # body of the kernel
    i = s.index("\n" + name + ":")
    j = s.index(".end_amdhsa_kernel", i)
    body = s[i:j].splitlines()
    n_sl = sum("scratch_load" in l for l in body)
    n_ss = sum("scratch_store" in l for l in body)
    # scratch accesses inside loops: lines annotated by the compiler with "in Loop:" headers are labels; approximate by
    # tracking the innermost label's Depth annotation
    depth, in_loop, mfma_depths = 0, [], []
    for l in body:
        m = re.search(r"Depth=(\d+)", l)
        if l.startswith(".LBB") or l.startswith("; %bb"):
            depth = int(m.group(1)) if m else 0
        if "scratch_load" in l or "scratch_store" in l:
            in_loop.append(depth)
        if "v_mfma" in l:
            mfma_depths.append(depth)
    # per loop depth: scratch accesses and MFMAs.  A persistent kernel's outermost loop (candidate blocks, depth 1) also holds
    # the prologue / tail code of a block, where a spill costs nothing (and, in the int8 sweep, the two float64 MFMAs of the
    # prologue's first K* tile); what matters is scratch traffic at the depths of the STEP loops -- the depths >= 2 that hold
    # the bulk of the MFMAs
    from collections import Counter
    sc, mf = Counter(in_loop), Counter(mfma_depths)
    step_depth = min((d for d in mf if mf[d] >= 8), default=None)
    in_step = sum(n for d, n in sc.items() if step_depth is not None and d >= step_depth)
    print(f"{dn[:100]:100s} vgpr={g('vgpr_count')} spillV={g('vgpr_spill_count')} spillS={g('sgpr_spill_count')} "
          f"scratch={g('private_segment_fixed_size')}B lds={g('group_segment_fixed_size')} "
          f"scratch ld/st={n_sl}/{n_ss} at loop depths {sorted(set(in_loop))} (deepest count {sum(d == max(in_loop) for d in in_loop) if in_loop else 0}); "
          f"per depth scratch {dict(sorted(sc.items()))} MFMA {dict(sorted(mf.items()))}: scratch accesses inside the step loops (depth >= {step_depth}) = {in_step}")
