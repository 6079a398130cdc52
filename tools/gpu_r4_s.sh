#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in new2 s1d1; do TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python tools/bitcheck_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/bitcheck_$v.txt; done
cmp $OUT/bitcheck_new2.txt $OUT/bitcheck_s1d1.txt && echo "bitcheck s1d1: identical to the committed kernel ($(wc -l < $OUT/bitcheck_s1d1.txt) configurations)" | tee -a $OUT/r04_defer2.txt
tail -4 $OUT/bitcheck_new2.txt
