#!/bin/bash
# GPU batch (round 6): the tall products' k-split on 128 x 128 tiles
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for N in 4096 8192 2048 1024; do
for cfg in "0 0 0" "1 0 0" "1 32 4096" "1 8 512"; do set -- $cfg
  echo -n "big=$1 "; TGP_KSPLIT_BIG=$1 TGP_KSPLIT_MAX=$2 TGP_KSPLIT_TARGET=$3 timeout 200 python tools/bench_ksplit.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-400
done; done | tee $OUT/r06_ksplit_big.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -x -q -m gpu -k "handful or gradient or joint or covariance or qei or greedy or fantas or penal" 2>&1 | tail -4
