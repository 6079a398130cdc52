#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for S in 0 16 32; do echo "== spare $S"; TGP_DAG_SPARE=$S timeout 120 python tools/bench_update_share.py 4096 2>&1 | grep -v amdgpu.ids; done
