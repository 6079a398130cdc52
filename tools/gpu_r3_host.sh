#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_dag.py tests/test_gpu_host.py -q -x -k "trial or optimi or fit or initiali" 2>&1 | tail -8
