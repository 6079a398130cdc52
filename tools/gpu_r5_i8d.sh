#!/bin/bash
# round 5, session D on the int8 sweep: second fragment's operand reads before the DMA; DMA position per half of the waves
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
for lib in tools/exp/libtgp_i8av1e0.so tools/exp/libtgp_i8av1.so tools/exp/libtgp_i8pos3.so tools/exp/libtgp_i8pos4.so; do
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_i8.py i8x4 i8x5 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/r05_i8_d.txt
