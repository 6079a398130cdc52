#!/bin/bash
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for B in 4 8; do echo "== burst $B N=6144"; TGP_DAG_BURST=$B timeout 100 python tools/bench_update.py 6144 2>&1 | grep update; done
for B in 8 12 16; do echo "== burst $B N=8192"; TGP_DAG_BURST=$B timeout 100 python tools/bench_update.py 8192 2>&1 | grep update; done
for B in 8 16; do echo "== burst $B N=12288"; TGP_DAG_BURST=$B timeout 100 python tools/bench_update.py 12288 2>&1 | grep update; done
