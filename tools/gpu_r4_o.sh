#!/bin/bash
# round 4: joint kernel, DMA descriptors of the next step computed behind the MFMA section (A/B) + parity tests
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in jl0 jl1 jl0 jl1; do
  echo "== $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 90 python -c "
import sys; sys.path.insert(0,'tools')
import bench_c4c5 as b; b.c4()" 2>&1 | grep 'C4 qEI\|rror' | cut -c1-200)"
done | tee $OUT/r04_joint_late.txt
TGP_LIB=$PWD/tools/exp/libtgp_jl1.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "joint or qei or c4 or covariance or greedy" 2>&1 | tail -4 | tee -a $OUT/r04_joint_late.txt
