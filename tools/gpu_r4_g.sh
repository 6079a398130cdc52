#!/bin/bash
# round 4, session G: i8 DMA-issue stagger A/B + the tests that cover the int8 sweeps and the new host-level AUTO tests
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( python tools/bench_i8.py i8x4 i8x5 auto
  TGP_LIB=$PWD/tools/exp/libtgp_st0.so python tools/bench_i8.py i8x4 i8x5 auto ) 2>&1 | grep -v amdgpu.ids | tee $OUT/r4g_i8_ab.txt
timeout 900 python -m pytest tests/test_gpu_i8.py tests/test_gpu_host.py tests/test_gpu_multi.py -q -x -k "i8x4_sweep or auto or group_under" 2>&1 | tail -6 | tee $OUT/r4g_tests.txt
# DAG tile-task DMA stagger A/B: update at 4096 / 8192 and the batched trials
for LIB in "" "$PWD/tools/exp/libtgp_ds0.so"; do
  echo "== lib: ${LIB:-default (stagger)}"
  TGP_LIB=$LIB python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids
  TGP_LIB=$LIB TGP_TIMING=1 python tools/bench_bo_step.py 4096 2>&1 | grep "N = 4096 exactly\|B=15 N=4096" | sort | uniq -c | sort -rn | head -4
done 2>&1 | tee $OUT/r4g_dag_ab.txt
timeout 600 python -m pytest tests/test_gpu_dag.py -q -x 2>&1 | tail -4 | tee -a $OUT/r4g_tests.txt
