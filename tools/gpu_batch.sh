#!/bin/bash
# GPU batch (round 6): qEI value-and-gradient on the device -- parity tests, timings
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "joint_forward or qei_value_and" 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/r06_qei_grad_tests2.txt
for a in "2048 5" "4096 5" "1024 10"; do timeout 600 python tools/bench_qei_grad.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-260; done | tee $OUT/r06_qei_grad2.txt
