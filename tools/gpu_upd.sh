#!/bin/bash
# update-path iteration: parity tests that touch the factorisation, then latency under the tuning knobs given as arguments
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "factor or update or append or clone or golden or posterior" 2>&1 | tail -3
for S in "$@"; do echo "== $S"; env $S python tools/bench_update.py 1024 2048 4096 8192 2>&1 | grep update; done
