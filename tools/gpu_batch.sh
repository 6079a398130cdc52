#!/bin/bash
# GPU batch (round 6): the persistent update kernel's split plan (T(i,i-2) / last burst of (i,i-1) as half-tile tasks)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dag.py -x -q 2>&1 | tail -5 | tee $OUT/r06_dag_split_tests.txt
{
for v in 256 0 256 0; do echo "== TGP_VARIANT=$v (256 = whole tiles, 0 = split plan)"; TGP_VARIANT=$v timeout 200 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids; done
for v in 256 0; do echo "== trace, TGP_VARIANT=$v"; TGP_VARIANT=$v TGP_DAG_TRACE=/tmp/dag_trace_$v.bin timeout 200 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | head -40; done
} 2>&1 | tee $OUT/r06_dag_split.txt
