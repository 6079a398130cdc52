#!/bin/bash
# find the crashing test: verbose names, per-file runs
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for f in tests/test_gpu_host.py tests/test_gpu_multi.py tests/test_gpu_dag.py tests/test_gpu_i8.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_c3.py; do
  echo "=== $f"; timeout 900 python -X faulthandler -m pytest $f -m gpu -q -v 2>&1 | grep -E "PASSED|FAILED|ERROR|Fatal|Segmentation|Abort|File \"/root|passed|failed|test_" | tail -12
done 2>&1 | tee $OUT/r4h_tests.txt | tail -120
