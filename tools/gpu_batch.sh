#!/bin/bash
# GPU batch (round 6, closing): suite + the driver's command with the final library (fixed-depth k-split slices, 16 / 2048 at every size)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for N in 4096 2048 1024; do timeout 200 python tools/bench_ksplit.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-400; done | tee $OUT/r06_ksplit_after.txt
bash tools/gpu_suite.sh r06p 2>&1 | tail -14
bash tools/gpu_evidence.sh r06p bench 2>&1 | tail -20
