// fused posterior sweep, kernel kind 1 (see tgp_kernels_sweep.inc; one TU per kind so the four compile in parallel)
#define TGP_SWEEP_KIND 1
#include "tgp_kernels_sweep.inc"
#include "tgp_kernels_sweep_dma.inc"
#include "tgp_kernels_joint.inc"
#include "tgp_kernels_sweep_i8.inc"
