"""The joint-mode kernel must compile without a single scratch access inside its step loops (CPU test: hipcc cross-compiles gfx950).

Why it is a test: at the 128-VGPR cap of a 1024-thread workgroup hipcc spills loop-invariant registers first, and every
reload of a spilled value is an ``s_waitcnt vmcnt(0)`` that also waits for the LDS-DMA in flight -- the round-2 form of
this kernel lost 12 % to exactly that (HISTORY.md section 0, profiles/r04_joint_knockout.txt).  A source change that
brings spills back is a performance regression no parity test sees."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_joint_kernel_has_no_scratch_access_in_its_step_loops(tmp_path):
    src = os.path.join(ROOT, "trieste_amd", "csrc", "tgp_kernels_sweep_k3.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-save-temps", "-c", src, "-o", "k3.o"],
                   cwd=tmp_path, check=True, capture_output=True, timeout=900)
    (asm,) = glob.glob(os.path.join(tmp_path, "*gfx950.s"))
    text = open(asm).read()
    meta = text[text.index("amdhsa.kernels:"):]
    seen = 0
    for block in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        if "joint_kernel" not in name:
            continue
        seen += 1
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        # Scratch accesses per loop depth (hipcc annotates every block label with the depth of the loop it sits in).  Depth 1 is
        # the persistent kernel's loop over candidate blocks: its prologue / tail code runs once per ~5 ms block and has no DMA
        # in flight, a spilled value there costs nothing (round 6's experiment with the block index drawn from a counter
        # left one such reload at d <= 4).  The STEP loops are at depth >= 2: none there, ever.
        depth, per_depth = 0, {}
        for line in body.splitlines():
            if line.startswith(".LBB") or line.startswith("; %bb"):
                m = re.search(r"Depth=(\d+)", line)
                depth = int(m.group(1)) if m else 0
            if "scratch_" in line:
                per_depth[depth] = per_depth.get(depth, 0) + 1
        assert all(d <= 1 for d in per_depth), f"{name}: scratch accesses per loop depth {per_depth}"
        assert sum(per_depth.values()) <= 4, f"{name}: scratch accesses per loop depth {per_depth}"
        assert "global_load_lds_dwordx4" in body, f"{name}: the LDS-DMA staging is gone"
    assert seen == 5, "one joint kernel per padded dimension 2, 4, 6, 8, 16"
