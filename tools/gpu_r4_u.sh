#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TGP_LIB=$PWD/tools/exp/libtgp_new2.so timeout 60 python tools/bitcheck_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/bitcheck_committed.txt
timeout 60 python tools/bitcheck_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/bitcheck_final.txt
cmp $OUT/bitcheck_committed.txt $OUT/bitcheck_final.txt && echo "final headline kernel: outputs bit-identical to the round-3 kernel on $(wc -l < $OUT/bitcheck_final.txt) configurations" | tee $OUT/r04_bitcheck.txt
tail -3 $OUT/bitcheck_final.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
