#!/bin/bash
# round 4: last k4-round of a step deferred across the barrier (joint kernel: df0 / df1; headline kernel: SGPR-base DMA s1,
# + deferred s1d1, against the committed kernel new2)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in df0 df1 df0 df1; do
  echo "== $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 90 python -c "
import sys; sys.path.insert(0,'tools')
import bench_c4c5 as b; b.c4()" 2>&1 | grep 'C4 qEI\|rror' | cut -c1-200)"
done | tee $OUT/r04_defer.txt
for v in new2 s1 s1d1 new2 s1 s1d1; do
  echo "== headline $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python bench.py --workload headline --no-cpu-baseline --no-acquire --no-secondary --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])")"
done | tee -a $OUT/r04_defer.txt
