#!/bin/bash
# round 5, session C on the int8 sweep: the lean step loop (SGPR-base DMA, rotating stage offsets) and where it issues its DMA
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
timeout 300 python -m pytest tests/test_gpu_i8.py -x -q -m gpu 2>&1 | tail -3
for lib in tools/exp/libtgp_base.so trieste_amd/libtgp.so tools/exp/libtgp_i8p0.so tools/exp/libtgp_i8p2.so tools/exp/libtgp_i8p1g.so; do
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_i8.py i8x4 i8x5 2>&1 | grep -v amdgpu.ids
done
TGP_LIB=$PWD/tools/exp/libtgp_i8p1tr.so timeout 120 python tools/i8_trace.py i8x4 2>&1 | grep -v amdgpu.ids | head -32
} | tee $OUT/r05_i8_c.txt
