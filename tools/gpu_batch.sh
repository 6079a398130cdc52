#!/bin/bash
# GPU batch (round 6): small joint calls through the skinny product -- the whole suite, timings
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06j 2>&1 | tail -30
for a in "2048 5" "4096 5"; do timeout 600 python tools/bench_qei_grad.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-260; done | tee $OUT/r06_joint_small.txt
