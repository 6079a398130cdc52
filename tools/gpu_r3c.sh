#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for n in ${1:-4096}; do timeout 200 python tools/dag_trace.py $n 2>&1 | grep -v amdgpu.ids > $OUT/dag_trace_$n.txt; done
echo "== dag tests"; timeout 900 python -m pytest tests/test_gpu_dag.py -m gpu -x -q 2>&1 | tail -3
