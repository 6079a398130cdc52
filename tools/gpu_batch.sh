#!/bin/bash
# GPU batch (round 6): greedy-batch acquisition latency at N = 2000 / 4000
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for N in 2000 4000; do echo "== N = $N"; timeout 400 python tools/bench_greedy.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-200; done | tee $OUT/r06_greedy.txt
