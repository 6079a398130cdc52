#!/bin/bash
# GPU batch (round 6, scratch script: the commands of the current A/B session)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_i8.py -x -q 2>&1 | grep -v "^$" | tail -25 > $OUT/r06_i8_tests.txt; cat $OUT/r06_i8_tests.txt
bash tools/gpu_ab.sh r06_i8_genmfma "python tools/bench_i8.py i8x4 i8x5 auto" tools/exp/libtgp_i8nomfma.so default tools/exp/libtgp_i8nomfma.so default
for v in i8trmfma; do
  echo "== $v i8x4"; TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 300 python tools/i8_trace.py i8x4 2>&1 | grep -v amdgpu.ids | head -40
done | tee $OUT/r06_i8_trace_mfma.txt
