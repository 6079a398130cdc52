#!/bin/bash
# GPU batch (round 6): the two-workgroup chain (and its plan) beyond NB = 47 (TGP_DAG_SPLIT_MAX_NB)
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for m in 48 127 48 127; do echo "== TGP_DAG_SPLIT_MAX_NB=$m (48 = default)"; TGP_DAG_SPLIT_MAX_NB=$m timeout 300 python tools/bench_update.py 5120 6144 7168 8192 2>&1 | grep -v amdgpu.ids | cut -c1-110; done | tee $OUT/r06_dag_duo_large.txt
