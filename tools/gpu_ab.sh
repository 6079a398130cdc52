#!/bin/bash
# A/B timings on ONE box (boxes differ by +- 3 %): a bench tool under several builds of the library.
#   usage: tools/gpu_ab.sh <out-name> "<command with its arguments>" <lib> [<lib> ...]      (lib: a path, or `default`)
#   e.g.   tools/gpu_ab.sh r05_i8_x "python tools/bench_i8.py i8x4 auto" default tools/exp/libtgp_x.so default tools/exp/libtgp_x.so
set -u; NAME=$1; CMD=$2; shift 2; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for lib in "$@"; do
  [ "$lib" = default ] && lib=trieste_amd/libtgp.so
  echo "== $lib"; TGP_LIB=$PWD/$lib timeout 300 $CMD 2>&1 | grep -v amdgpu.ids
done | tee $OUT/$NAME.txt
