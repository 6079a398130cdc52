#!/bin/bash
# round 5, session H on the int8 sweep: block-step loop with three inlined step bodies (scalar bookkeeping out of the step)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
for lib in tools/exp/libtgp_i8av1.so trieste_amd/libtgp.so; do
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_i8.py i8x4 i8x5 auto 2>&1 | grep -v amdgpu.ids
done
timeout 400 python -m pytest tests/test_gpu_i8.py -x -q -m gpu 2>&1 | tail -3
TGP_LIB=$PWD/tools/exp/libtgp_i8tr2.so timeout 120 python tools/i8_trace.py i8x4 2>&1 | grep -v amdgpu.ids | head -32
} | tee $OUT/r05_i8_h.txt
