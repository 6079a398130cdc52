#!/bin/bash
# GPU batch (round 6, closing): the --runslow twins and the driver's command with the final library
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --runslow only 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/r06_gpu_tests_runslow.txt
bash tools/gpu_evidence.sh r06h bench 2>&1 | tail -20
for a in "2048 5"; do timeout 600 python tools/bench_qei_grad.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-300; done | tee $OUT/r06_joint_small2.txt
