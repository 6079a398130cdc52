#!/bin/bash
# GPU batch (round 6): a default acquire with the lock-step L-BFGS-B driver against the scipy.optimize.minimize + greenlet form (TGP_LOCKSTEP=0)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/r06_lockstep_tests.txt
{
for N in 1024 4096; do for m in 0 1 0 1; do
  echo "== N = $N, lock-step driver: $m"
  TGP_LOCKSTEP=$m timeout 300 python tools/prof_acquire.py $N 2>&1 | grep "acquire_single ms"
done; done
} | tee $OUT/r06_lockstep_acquire.txt
