#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tile map 1024"; timeout 120 python tools/dag_debug.py 1024 2>&1 | grep -v amdgpu.ids | tail -24
echo "== dag tests"; timeout 900 python -m pytest tests/test_gpu_dag.py -m gpu -x -q 2>&1 | tail -12
echo "== update latency (DAG)"; timeout 300 python tools/bench_update.py 512 1024 2048 4096 8192 2>&1 | grep update
