#!/bin/bash
# Development aid: experimental builds of libtgp's sweep TUs with extra -D flags (timing knock-outs give WRONG results).
# usage: tools/build_exp.sh <tag> "<flags>" [<tag> "<flags>" ...]   -> tools/exp/libtgp_<tag>.so   (use with TGP_LIB=...)
#        e.g. tools/build_exp.sh x1 "-DTGP_EXP=1" vol "-DTGP_DMA_VOLATILE_READS=1"
set -e
cd "$(dirname "$0")/../trieste_amd/csrc"
mkdir -p ../../tools/exp
while [ $# -ge 2 ]; do
  TAG=$1; FLAGS=$2; shift 2
  B=/tmp/tgp_exp_$TAG; mkdir -p $B
  for k in 0 1 2 3; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c tgp_kernels_sweep_k$k.hip -o $B/k$k.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libtgp_$TAG.so tgp_api.o tgp_group.o tgp_kernels_linalg.o tgp_kernels_leaf.o tgp_kernels_dag.o tgp_kernels_misc.o tgp_kernels_grad.o tgp_kernels_traj.o $B/k0.o $B/k1.o $B/k2.o $B/k3.o -ldl -lpthread
  echo built $TAG
done
