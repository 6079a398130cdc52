#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for x in 0.5 1.1 1.6; do echo "== xshift $x"; TGP_DAG_XSHIFT=$x timeout 60 python tools/dag_trace.py 4096 2>&1 | grep -v amdgpu.ids | head -2; TGP_DAG_XSHIFT=$x timeout 90 python tools/bench_update.py 4096 8192 2>&1 | grep update; done
