#!/bin/bash
# round 4, GPU session B: i8 kernel A/B (error model / r03 kernel on the same box), the tests session A did not reach,
# the batched trial evaluations, the fit timing
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( python tools/bench_i8.py i8x4 i8x5 auto f64
  TGP_LIB=$PWD/tools/exp/libtgp_e0.so python tools/bench_i8.py i8x4 i8x5
  TGP_LIB=$PWD/tools/exp/libtgp_r03k.so python tools/bench_i8.py i8x4 i8x5 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/r4b_i8_ab.txt
timeout 900 python -m pytest tests/test_gpu_dag.py -q -x -k "batched" 2>&1 | tail -15 | tee $OUT/r4b_batch_tests.txt
timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/r4b_bo_step.txt
timeout 1200 python -m pytest tests/test_gpu_i8.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -s 2>&1 | grep -v "^$" | tail -40 > $OUT/r4b_tests.txt; tail -30 $OUT/r4b_tests.txt
