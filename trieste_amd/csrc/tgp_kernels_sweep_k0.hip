// sweep kernels for kernel kind 0 (see tgp_kernels_sweep*.inc)
#define TGP_SWEEP_KIND 0
#include "tgp_kernels_sweep.inc"
#include "tgp_kernels_sweep_ws.inc"
#include "tgp_kernels_sweep_u16.inc"
