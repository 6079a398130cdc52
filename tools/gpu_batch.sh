#!/bin/bash
# GPU batch (round 6): ordinary (L2-cacheable) loads for the write-once operand tiles of the persistent update kernel's tile tasks -- tests, then A/B on one box
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/r06_dag_plain_tests.txt
{
for lib in tools/exp/libtgp_plainl0.so trieste_amd/libtgp.so tools/exp/libtgp_plainl0.so trieste_amd/libtgp.so; do
  echo "== $lib (plainl0 = every operand past the L2; libtgp = ordinary loads for L and W's diagonal tiles)"
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids
done
echo "== batched fit, plainl0 then libtgp"
for lib in tools/exp/libtgp_plainl0.so trieste_amd/libtgp.so; do
  TGP_LIB=$PWD/$lib timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids | grep "find_best\|COLD"
done
TGP_DAG_TRACE=/tmp/dag_trace.bin timeout 100 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -6
TGP_DAG_TRACE=/tmp/dag_trace.bin timeout 100 python tools/dag_trace.py 8192 2>&1 | grep -v amdgpu.ids | cut -c1-200 | grep -v "<-\|task \|step [0-9]" | head -20
} | tee $OUT/r06_dag_plain.txt
