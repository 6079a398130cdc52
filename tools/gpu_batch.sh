#!/bin/bash
# GPU batch (round 6): the two-workgroup chain of the persistent update kernel -- tests, then update timing and traces (variant 512 = one chain)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/r06_dag_duo_tests.txt
{
for v in 512 0 512 0; do
  echo "== TGP_VARIANT=$v (512 = one chain, 0 = two-workgroup chain)"
  TGP_VARIANT=$v timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids
done
for v in 512 0; do
  echo "== trace, TGP_VARIANT=$v"
  TGP_VARIANT=$v TGP_DAG_TRACE=/tmp/dag_trace.bin timeout 100 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-400 | head -40
done
} | tee $OUT/r06_dag_duo.txt
