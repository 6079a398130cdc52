#!/bin/bash
# round 5: the canary of TGP_PREC_AUTO -- tests, and what it costs on the headline sweep
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_i8.py -x -q -m gpu -s 2>&1 | grep "\[margin\] auto canary\|passed\|failed\|Error\|assert" | tee $OUT/r05_canary_tests.txt
{
echo "# tools/bench_i8.py, one box: plain i8x4 (no bounds, no repair, no canary) against AUTO (bounds + repair + canary)"
for i in 1 2; do timeout 100 python tools/bench_i8.py i8x4 auto i8x5 f64 2>&1 | grep -v amdgpu.ids; done
} | tee $OUT/r05_i8_ab.txt
