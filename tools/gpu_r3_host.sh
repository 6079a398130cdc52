#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/bench_update_share.py 4096 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_bo_step.py 1000 2>&1 | grep -v amdgpu.ids
