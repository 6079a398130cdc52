#!/bin/bash
# A/B of the persistent `update` kernel on ONE box: tests/test_gpu_dag.py under the default library, then update latency
# (N = 4096 / 8192) and the batched fit timings under every library given.
#   usage: tools/gpu_dag_ab.sh <out-name> <lib> [<lib> ...]      (lib: a path, or `default`)
set -u; NAME=$1; shift; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 250 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/${NAME}_tests.txt
for lib in "$@"; do
  [ "$lib" = default ] && lib=trieste_amd/libtgp.so
  echo "== $lib"
  TGP_LIB=$PWD/$lib timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids
  TGP_LIB=$PWD/$lib timeout 200 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids | grep "find_best\|COLD"
done | tee $OUT/$NAME.txt
