#!/bin/bash
# round 5, session G on the int8 sweep: cache policy of the K* slab stream (the slabs thrash the L2 the W tiles live in)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
for lib in trieste_amd/libtgp.so tools/exp/libtgp_i8n1.so tools/exp/libtgp_i8n2.so tools/exp/libtgp_i8n3.so tools/exp/libtgp_i8n4.so tools/exp/libtgp_i8n5.so tools/exp/libtgp_i8n1s.so; do
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/r05_i8_g.txt
