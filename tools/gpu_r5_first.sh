#!/bin/bash
# Prepared at the end of round 4 (no GPU budget left) -- the first GPU session of the next round, ~4 GPU-minutes.
# BEFORE calling gpurun, on the CPU side:   make -C trieste_amd/csrc -j8
#                                           tools/build_exp_tu.sh dagk tgp_kernels_dag "-DTGP_DAG_KERNARG=1"
#                                           tools/build_exp_tu.sh dagc tgp_kernels_dag "-DTGP_DAG_CHAIN_LOCAL=1"
#                                           tools/build_exp_tu.sh dagkc tgp_kernels_dag "-DTGP_DAG_KERNARG=1 -DTGP_DAG_CHAIN_LOCAL=1"
# Measures the two variants that were written but never run (csrc/tgp_kernels_dag.hip): TGP_DAG_KERNARG -- the persistent
# kernel's workers read their launch arguments from the kernarg segment with scalar loads instead of from a scratch copy
# (throughput-bound cases: N = 8192, batched fits); TGP_DAG_CHAIN_LOCAL -- the chain works on a wave-uniform local copy of
# the arguments instead of re-loading fields from scratch on its critical path (chain-bound case: N = 4096 update).
#   1. correctness of the variant: tests/test_gpu_dag.py (bit-identity against the recursion, NOT_PD, batched members)
#   2. timing, shipped against variant: update at N = 4096 / 8192 (tools/bench_update.py), the batched fit
#      (tools/bench_bo_step.py: find_best_model_initialization(90), cold optimize)
# Turn the macro on only if (1) is green and (2) is faster; HISTORY.md section 0 has the ISA facts behind it.
set -u; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$PWD/tools/exp/libtgp_dagkc.so
[ -f $V ] || { echo "build the tools/exp/libtgp_dag*.so variants first (see the header)"; exit 1; }
TGP_LIB=$V timeout 200 python -m pytest tests/test_gpu_dag.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/r05_dagk_tests.txt
for lib in "" $PWD/tools/exp/libtgp_dagk.so $PWD/tools/exp/libtgp_dagc.so $V; do
  echo "== ${lib:-shipped}"
  TGP_LIB=${lib:-$PWD/trieste_amd/libtgp.so} timeout 100 python tools/bench_update.py 4096 8192 2>&1 | grep -v amdgpu.ids
  TGP_LIB=${lib:-$PWD/trieste_amd/libtgp.so} timeout 200 python tools/bench_bo_step.py 4096 2>&1 | grep -v amdgpu.ids | grep "find_best\|COLD"
done | tee $OUT/r05_dagk_ab.txt
