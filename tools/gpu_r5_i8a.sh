#!/bin/bash
# round 5, session A on the int8 sweep: per-phase shader-clock trace of a step + knock-out timings of the shipped kernel.
# before: tools/build_exp.sh i8tr "-DTGP_I8_TRACE=1"; for v in 1 2 4 8 9 11 15: tools/build_exp_tu.sh i8ko$v tgp_kernels_sweep_k3 "-DTGP_I8_KO=$v"
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
TGP_LIB=$PWD/tools/exp/libtgp_i8tr.so timeout 120 python tools/i8_trace.py i8x4 2>&1 | grep -v amdgpu.ids
timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 8 9 11 15; do
  TGP_LIB=$PWD/tools/exp/libtgp_i8ko$v.so timeout 100 python tools/bench_i8.py i8x4 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/r05_i8_trace_ko.txt
