#!/bin/bash
# round 4, GPU session E: batched trial evaluations, enqueue-only groups + jointly simulated dispatch list
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dag.py -q -x -k "batched or trial" 2>&1 | tail -5 | tee $OUT/r4e_tests.txt
timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep -v "amdgpu.ids" | tee $OUT/r4e_bo_step.txt
TGP_TIMING=1 timeout 300 python tools/bench_bo_step.py 4096 2>&1 | grep "trial batch" | sort | uniq -c | sort -rn | head -8 | tee $OUT/r4e_batch_timing.txt
