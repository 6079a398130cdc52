#!/bin/bash
# GPU batch (round 6, closing): smoke() with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r06_smoke.txt 2>&1; echo "rc=$?" >> $OUT/r06_smoke.txt
grep -v "amdgpu.ids\|Hostname\|Librccl" $OUT/r06_smoke.txt | tail -8
