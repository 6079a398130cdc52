#!/bin/bash
# GPU batch (round 6): the two-workgroup chain of the persistent update kernel -- tests, timing, trace
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/r06_dag_duo_tests.txt
{
for v in 512 0 512 0; do
  echo "== TGP_VARIANT=$v (512 = one chain, 0 = two-workgroup chain)"
  TGP_VARIANT=$v timeout 100 python tools/bench_update.py 4096 2>&1 | grep -v amdgpu.ids
done
TGP_VARIANT=0 TGP_DAG_TRACE=/tmp/dag_trace.bin timeout 100 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -8
} | tee $OUT/r06_dag_duo.txt
