#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host.py -x -q -m gpu -k "refines_a_joint" --durations=3 2>&1 | grep -v amdgpu.ids | tail -25 | tee $OUT/r06_qei_ego_test.txt
