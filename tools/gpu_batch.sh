#!/bin/bash
# GPU batch (round 6): cross-barrier deferral in the persistent update kernel's tile task -- tests, then A/B on one box (defer0 = without)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/r06_dag_defer_tests.txt
{
for lib in tools/exp/libtgp_defer0.so trieste_amd/libtgp.so tools/exp/libtgp_defer0.so trieste_amd/libtgp.so; do
  echo "== $lib"
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_update.py 2048 4096 8192 2>&1 | grep -v amdgpu.ids | cut -c1-100
done
for lib in tools/exp/libtgp_defer0.so trieste_amd/libtgp.so; do
  echo "== fit, $lib"
  TGP_LIB=$PWD/$lib timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids | grep "batched at\|COLD"
done
} | tee $OUT/r06_dag_defer.txt
