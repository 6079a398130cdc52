#!/bin/bash
# GPU batch (round 6): qEI value-and-gradient -- new parity tests, timings; then the whole suite and the closing bench line
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "joint_forward or qei_value_and or handful" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/r06_qei_grad_tests.txt
for a in "2048 5" "2048 3" "4096 5" "1024 10"; do timeout 600 python tools/bench_qei_grad.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-260; done | tee $OUT/r06_qei_grad.txt
bash tools/gpu_suite.sh r06i 2>&1 | tail -8
bash tools/gpu_evidence.sh r06g bench 2>&1 | tail -20
