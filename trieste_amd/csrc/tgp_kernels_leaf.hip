// 128 x 128 leaf of the factor recursion: Cholesky factor L of a diagonal block AND W = L^-1 in ONE workgroup.
//
// Replaces, per 128 rows, two 64-leaves and the three products of their parent node (five dependent launches, each
// with the ~4.7 us floor of a dependent kernel on this part: profiles/r02_update_breakdown.txt).  Reference call
// site: tf.linalg.cholesky(K + s I) inside gpflow's GPR posterior, reached from
// trieste/models/gpflow/models.py:171-186 -> interface.py:108-112.
//
// The serial part of a Cholesky factorisation is its pivot chain; everything here is arranged so that a pivot costs
// a wave a couple of dozen instructions, NO memory round trip and NO scalar-register traffic:
//   * 16-column panels, one matrix row per lane, the 16 entries of the row in registers.  Every 16-lane DPP row of
//     a panel wave also carries a replica of the 16 x 16 diagonal block (lane l: row l & 15), so the pivot and the
//     15 - j multipliers of step j reach every lane through the DPP operand of the instruction that uses them:
//         v_fmac_f64_dpp  x[k], -l_diag, l_own  row_newbcast:k      (x[k] -= L[k][j] * L[row][j])
//     -- one instruction per updated entry, no v_readlane, no LDS, no barrier, no wait inside the 16 steps.
//     Rows below the diagonal block ride on the same instructions: the panel solve  L_p = A_p L_d^-T  costs
//     nothing extra, and sixteen identity rows riding along come out as  L_d^-T,  the inverse of the diagonal block.
//     Two waves (on two SIMDs) hold the at most 112 + 16 such rows.
//   * Trailing update  S -= L_p L_p^T  on v_mfma_f64_16x16x4_f64, all eight waves.
//   * The inverse is built right-looking and IN PLACE: once block column kb of the trailing matrix has become L
//     (written to global memory by the panel waves), its LDS slots are dead and receive block column kb of
//     T = E_kb ... E_0 (E_k: the block elimination step of column k), which after the last panel is W.  One
//     128 x 128 array holds both.  All of this except one block column of the trailing update is done by the six
//     other waves WHILE the two panel waves work on the next panel (disjoint slots: see the schedule at the kernel).
// A non-positive (or NaN) pivot is not patched: it turns the rest of the block into NaN and is reported through
// `info` (1 + global index of the first one), which is all the caller looks at.
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

// development aid (tools/ubench_leaf.hip): time stamps of thread 0 at the phase boundaries
#ifndef TGP_LEAF_TICK
#define TGP_LEAF_TICK(i)
#define TGP_LEAF_TICKW(i)
#endif

namespace tgp {
namespace {

#include "tgp_leaf_dev.inc"

__global__ __launch_bounds__(512) void leaf128_kernel(const double* __restrict__ A, double* __restrict__ L,
                                                      double* __restrict__ W, int64_t ld, int64_t off,
                                                      int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double S[];  // [QN][QS], then the work-item table
  WorkItem* const items = (WorkItem*)(S + QN * QS);           // [QB][NWORK][WSLOTS]
  const int tid = threadIdx.x;
  {  // rows of the lower triangle, two entries per thread (coalesced), all 16 loads of a thread in flight together
    v2d x[QN * QN / 1024];
#pragma unroll
    for (int it = 0; it < QN * QN / 1024; ++it) {
      const int e = tid + 512 * it, i = e >> 6, j = (e & 63) * 2;
      x[it] = (v2d){0.0, 0.0};
      if (j <= i) x[it] = *(const v2d*)(A + (off + i) * ld + off + j);
      if (j + 1 > i) x[it].y = 0.0;
    }
#pragma unroll
    for (int it = 0; it < QN * QN / 1024; ++it) {
      const int e = tid + 512 * it, i = e >> 6, j = (e & 63) * 2;
      *(v2d*)(S + i * QS + j) = x[it];
    }
  }
  leaf_build_items(items, tid);
  lds_sync_t* const sync = (lds_sync_t*)(__attribute__((address_space(3))) char*)(items + QB * NWORK * WSLOTS);  // one word behind the table
  if (tid == 0) *sync = 0;
  __syncthreads();
  TGP_LEAF_TICK(0);
  leaf_core<false>(S, items, L, ld, off, info, sync);
  // W = T: the 36 blocks on and below the block diagonal (everything above stays zero: tgp_api.hip keeps the upper
  // triangles of L and W zeroed); 2 entries per thread and pass
  {
    constexpr int NPASS = QN * QN / 1024;
    v2d x[NPASS];
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int e = tid + 512 * it, i = e >> 6, j = (e & 63) * 2;
      x[it] = *(const v2d*)(S + i * QS + j);
    }
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int e = tid + 512 * it, i = e >> 6, j = (e & 63) * 2;
      if ((j >> 4) <= (i >> 4)) *(v2d*)(W + (off + i) * ld + off + j) = x[it];
    }
  }
  TGP_LEAF_TICK(33);
}

}  // namespace

void launch_leaf128(hipStream_t s, const double* A, double* L, double* W, int64_t ld, int64_t off, int* info) {
  constexpr size_t shmem = (size_t)QN * QS * sizeof(double) + QB * NWORK * WSLOTS * sizeof(WorkItem) + 64;
  // > 64 KiB of dynamic LDS is opt-in, per function and per device: set on every launch
  (void)hipFuncSetAttribute((const void*)leaf128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(leaf128_kernel, dim3(1), dim3(512), shmem, s, A, L, W, ld, off, info);
}

}  // namespace tgp
