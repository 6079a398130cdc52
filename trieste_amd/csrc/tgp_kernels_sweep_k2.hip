// sweep kernels for kernel kind 2 (see tgp_kernels_sweep.inc / tgp_kernels_sweep_ws.inc)
#define TGP_SWEEP_KIND 2
#include "tgp_kernels_sweep.inc"
#include "tgp_kernels_sweep_ws.inc"
