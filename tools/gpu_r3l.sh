#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/bench_update.py 4096 8192 2>&1 | grep update
timeout 400 python -m pytest tests/test_gpu_dag.py -q -x 2>&1 | tail -3
