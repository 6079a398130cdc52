#!/bin/bash
# Prepared at the end of round 4 (no GPU budget left) -- the first GPU session of the next round, ~4 GPU-minutes.
# BEFORE calling gpurun, on the CPU side:   make -C trieste_amd/csrc -j8
#                                           tools/build_exp_tu.sh dagk tgp_kernels_dag "-DTGP_DAG_KERNARG=1"
# Measures the one variant that was written but never run: the persistent kernel's workers reading their launch arguments
# from the kernarg segment with scalar loads instead of from a scratch copy (csrc/tgp_kernels_dag.hip, TGP_DAG_KERNARG).
#   1. correctness of the variant: tests/test_gpu_dag.py (bit-identity against the recursion, NOT_PD, batched members)
#   2. timing, shipped against variant: update at N = 4096 / 8192 (tools/bench_update.py), the batched fit
#      (tools/bench_bo_step.py: find_best_model_initialization(90), cold optimize)
# Turn the macro on only if (1) is green and (2) is faster; HISTORY.md section 0 has the ISA facts behind it.
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/tools/exp/libtgp_dagk.so
[ -f $V ] || { echo "build tools/exp/libtgp_dagk.so first (see the header)"; exit 1; }
TGP_LIB=$V timeout 200 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/r05_dagk_tests.txt
for lib in "" $V; do
  echo "== ${lib:-shipped}"
  TGP_LIB=${lib:-$PWD/trieste_amd/libtgp.so} timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids
  TGP_LIB=${lib:-$PWD/trieste_amd/libtgp.so} timeout 200 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids | grep "find_best\|COLD"
done | tee $OUT/r05_dagk_ab.txt
