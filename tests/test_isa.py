"""The joint-mode kernel must compile without a single scratch access (CPU test: hipcc cross-compiles gfx950).

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
def test_joint_kernel_has_no_scratch_access(tmp_path):
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
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
        spilled = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1))
        assert (scratch, spilled) == (0, 0), f"{name}: {spilled} spilled VGPRs, {scratch} bytes of scratch"
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        assert "scratch_" not in body, name
        assert "global_load_lds_dwordx4" in body, f"{name}: the LDS-DMA staging is gone"
    assert seen == 5, "one joint kernel per padded dimension 2, 4, 6, 8, 16"
