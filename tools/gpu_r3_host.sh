#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c3.py tests/test_gpu_host.py -q -x 2>&1 | tail -6
