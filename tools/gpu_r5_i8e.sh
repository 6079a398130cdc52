#!/bin/bash
# round 5, session E on the int8 sweep: knock-outs of the lean kernel (timing only)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
TGP_LIB=$PWD/trieste_amd/libtgp.so timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 8 16 5 7 15; do
  TGP_LIB=$PWD/tools/exp/libtgp_i8k$v.so timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/r05_i8_e.txt
