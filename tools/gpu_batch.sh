#!/bin/bash
# GPU batch (round 6): generation first on the light waves WITH the heavy waves holding priority 3 through their MFMA phase
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_ab.sh r06_i8_gfprio "python tools/bench_i8.py i8x4 i8x5" default tools/exp/libtgp_gfp.so default tools/exp/libtgp_gfp.so
echo "== trace gfp"; TGP_LIB=$PWD/tools/exp/libtgp_gfptr.so timeout 300 python tools/i8_trace.py i8x4 2>&1 | grep -v amdgpu.ids | head -30 | tee $OUT/r06_i8_trace_gfprio.txt
