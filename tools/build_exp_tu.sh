#!/bin/bash
# Development aid: an experimental build of libtgp with ONE translation unit rebuilt with extra -D flags (the others are the
# objects of the regular build: run `make -C trieste_amd/csrc` first).
# usage: tools/build_exp_tu.sh <tag> <tu> "<flags>"   -> tools/exp/libtgp_<tag>.so   (use with TGP_LIB=...)
#        e.g. tools/build_exp_tu.sh dagk tgp_kernels_dag "-DTGP_DAG_KERNARG=1"
set -e
cd "$(dirname "$0")/../trieste_amd/csrc"
mkdir -p ../../tools/exp
TAG=$1; TU=$2; FLAGS=$3
B=/tmp/tgp_exp_$TAG; mkdir -p $B
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $FLAGS -c $TU.hip -o $B/$TU.o
OBJS=""
for o in tgp_api tgp_group tgp_kernels_linalg tgp_kernels_leaf tgp_kernels_dag tgp_kernels_misc tgp_kernels_grad tgp_kernels_traj tgp_kernels_sweep_k0 tgp_kernels_sweep_k1 tgp_kernels_sweep_k2 tgp_kernels_sweep_k3; do
  if [ "$o" = "$TU" ]; then OBJS="$OBJS $B/$o.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libtgp_$TAG.so $OBJS -ldl -lpthread
echo built $TAG
