#!/bin/bash
# GPU batch (round 6): the row-group split of small sweeps at N <= 2048 -- parity tests, then EGO's initial sweep and a default acquire
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_c3.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/r06_split_small_tests.txt
{
for N in 512 1024 2048 3072 4096; do echo "== N = $N"; timeout 300 python tools/bench_acquire.py $N 2>&1 | grep -v amdgpu.ids | grep "acq_topk\|acq_argmax" | tr '\n' ';'; echo; timeout 300 python tools/prof_acquire.py $N 2>&1 | grep "acquire_single ms"; done
} | tee $OUT/r06_split_small.txt
