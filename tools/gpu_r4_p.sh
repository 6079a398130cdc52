#!/bin/bash
# round 4: where the Gram phase of the rebuilt joint kernel spends its time (knock-outs: WRONG results, timing only)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in new2 k512 k640 k128; do
  echo "== $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 90 python -c "
import sys; sys.path.insert(0,'tools')
import bench_c4c5 as b; b.c4()" 2>&1 | grep 'C4 qEI\|rror' | cut -c1-200)"
done | tee $OUT/r04_joint_gram_ko.txt
