#!/bin/bash
set -u; export TMPDIR=/tmp
for LIB in "" dk1 dk5 dpr ds0; do
  L=""; [ -n "$LIB" ] && L=$PWD/tools/exp/libtgp_$LIB.so
  echo "== lib: ${LIB:-default (stagger, late issue after k4 = 3)}"
  TGP_LIB=$L python tools/bench_update.py 4096 8192 8192 2>&1 | grep -v amdgpu.ids | cut -c1-60
  TGP_LIB=$L TGP_TIMING=1 python tools/bench_cold_fit.py 4096 2>&1 | grep "B=15 N=4096" | awk '{print $11}' | sort -n | head -3 | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/r4k_dag_ab.txt
