#!/bin/bash
# GPU batch (round 6): host profile of a cold optimize()
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for N in 1024; do echo "== N = $N"; timeout 300 python tools/prof_optimize.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-150 | head -48; done | tee $OUT/r06_prof_optimize.txt
