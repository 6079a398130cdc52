#!/bin/bash
# GPU batch (round 6): launch groups of up to 45 members for the batched prior draws below N = 4096 (TGP_TRIAL_BATCH_MAX=16: the old groups of 15)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dag.py -x -q -m gpu -k "batched or trial" 2>&1 | tail -4 | tee $OUT/r06_trial_groups_tests.txt
{
for N in 512 1024 2048 3072 3840; do
  for m in 16 0 16 0; do
    echo "== N = $N, TGP_TRIAL_BATCH_MAX=$m (0 = default: 45 below N = 4096)"
    TGP_TRIAL_BATCH_MAX=$m timeout 300 python tools/fit_small_probe.py $N 2>&1 | grep -v amdgpu.ids | grep "find_best\|optimize" | tail -3
  done
done
} | tee $OUT/r06_trial_groups.txt
