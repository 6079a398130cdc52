#!/bin/bash
# GPU batch (round 6): the two-pass gradient tail -- the whole suite, then a default acquire
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06e 2>&1 | tail -5
{
for N in 1024 4096; do echo "== N = $N"; timeout 300 python tools/bench_acquire.py $N 2>&1 | grep -v amdgpu.ids | grep "acq_value_grad"; timeout 300 python tools/prof_acquire.py $N 2>&1 | grep "acquire_single ms"; done
} | tee $OUT/r06_grad_tail.txt
