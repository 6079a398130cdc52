#!/bin/bash
# round 5, session F on the int8 sweep: what bounds the step when the MFMAs are gone (timing only)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
for v in 12 14 44 28 46 36; do
  TGP_LIB=$PWD/tools/exp/libtgp_i8k$v.so timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/r05_i8_f.txt
