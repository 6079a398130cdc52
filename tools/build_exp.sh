#!/bin/bash
# Development aid: timing-experiment builds of libtgp (kernels with parts of the sweep's hot loop removed; WRONG results).
# usage: tools/build_exp.sh <bits> ...   -> tools/exp/libtgp_x<bits>.so   (use with TGP_LIB=...)
set -e
cd "$(dirname "$0")/../trieste_amd/csrc"
mkdir -p ../../tools/exp
for X in "$@"; do
  B=/tmp/tgp_exp_$X; mkdir -p $B
  for k in 0 1 2 3; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTGP_EXP=$X -c tgp_kernels_sweep_k$k.hip -o $B/k$k.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libtgp_x$X.so tgp_api.o tgp_group.o tgp_kernels_linalg.o tgp_kernels_misc.o tgp_kernels_grad.o tgp_kernels_traj.o $B/k0.o $B/k1.o $B/k2.o $B/k3.o -ldl -lpthread
  echo built x$X
done
