#!/bin/bash
# round 5, session B on the int8 sweep: plain loads instead of LDS-DMA for the W tiles (issue-cost probe), DMA after the MFMAs
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
for v in 16 32 64; do
  TGP_LIB=$PWD/tools/exp/libtgp_i8ko$v.so timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
done
TGP_LIB=$PWD/tools/exp/libtgp_i8tr32.so timeout 120 python tools/i8_trace.py i8x4 2>&1 | grep -v amdgpu.ids | head -32
} | tee $OUT/r05_i8_b.txt
