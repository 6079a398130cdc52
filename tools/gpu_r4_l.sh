#!/bin/bash
# round 4: knock-out study of joint_kernel at C4 (timing only; the ko* builds give WRONG results) + Gram-phase variants
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in base ko1 ko2 ko19 ko4 ko8 ko16 ko63 g5 g4l2 g5l2 base; do
  echo "== $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python -c "
import sys; sys.path.insert(0,'tools')
import bench_c4c5 as b; b.c4()" 2>&1 | grep 'C4 qEI' | cut -c1-200)"
done | tee $OUT/r04_joint_ko.txt
