#!/bin/bash
# round 4: headline kernel with the last k4-round deferred across the barrier: timing of two forms against the committed
# kernel, and checksums of the sweep outputs under each build (must be identical)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in new2 d1l1 s1d1 d1l1 s1d1; do
  echo "== headline $v: $(TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python bench.py --workload headline --no-cpu-baseline --no-acquire --no-secondary --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])")"
done | tee $OUT/r04_defer2.txt
for v in new2 d1l1 s1d1; do TGP_LIB=$PWD/tools/exp/libtgp_$v.so timeout 120 python tools/bitcheck_sweep.py 2>/dev/null > $OUT/bitcheck_$v.txt; done
for v in d1l1 s1d1; do cmp $OUT/bitcheck_new2.txt $OUT/bitcheck_$v.txt && echo "bitcheck $v: identical to the committed kernel ($(wc -l < $OUT/bitcheck_$v.txt) configurations)"; done | tee -a $OUT/r04_defer2.txt
head -3 $OUT/bitcheck_new2.txt
