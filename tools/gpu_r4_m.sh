#!/bin/bash
# round 4: joint_kernel rebuilt (LDS-DMA main loop, symmetric DMA-fed Gram phase): timing against the old kernel
# (tools/exp/libtgp_base.so) and the parity tests that go through it
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in base new new2 new new2; do
  echo "== $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 90 python -c "
import sys; sys.path.insert(0,'tools')
import bench_c4c5 as b; b.c4()" 2>&1 | grep 'C4 qEI\|rror' | cut -c1-200)"
done | tee $OUT/r04_joint_new.txt
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "joint or qei or c4 or covariance or greedy" 2>&1 | tail -8 | tee -a $OUT/r04_joint_new.txt
