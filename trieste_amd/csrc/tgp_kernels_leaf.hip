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

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int QN = 128, QS = QN + 4, QB = QN / 16;  // block size, LDS row stride (doubles), 16-blocks per side

// acc -= (lane K of this lane's 16-lane row: src) * y.  `src` must have been written >= 2 wait states earlier
// (VALU write -> DPP read hazard: the compiler does not see inside the asm; callers pass `src` through dpp_ready).
template <int K>
__device__ __forceinline__ void fmac_nbcast(double& acc, double src, double y) {
  asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(y), "n"(K));
}
template <int K>
__device__ __forceinline__ double mov_nbcast(double src) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
  return r;
}
__device__ __forceinline__ void dpp_ready(double& x) { asm("s_nop 1" : "+v"(x)); }

// 1 / sqrt(p): v_rsq_f64 (~2^-26) and one coupled Newton step (~2 ulp); NaN for p < 0 and for NaN, +inf for p = 0
__device__ __forceinline__ double rsqrt_short(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double g = p * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  const double h2 = fma(h, r, h);
  return h2 + h2;
}

template <int J, int K>
struct PanelUpdate {
  static __device__ __forceinline__ void run(double (&d)[16], double (&a)[16], double ld, double la) {
    if constexpr (K < 16) {
      fmac_nbcast<K>(d[K], ld, ld);
      fmac_nbcast<K>(a[K], ld, la);
      PanelUpdate<J, K + 1>::run(d, a, ld, la);
    }
  }
};
// step J of the panel: p = pivot (already broadcast).  The pivot of step J + 1 is fetched as soon as its entry has
// been updated, so that its 1/sqrt chain runs underneath the remaining updates of step J.
template <int J>
struct PanelStep {
  static __device__ __forceinline__ void run(double (&d)[16], double (&a)[16], double p) {
    const double rs = rsqrt_short(p);
    double ld = d[J] * rs;  // column J of L: the diagonal block's rows ...
    const double la = a[J] * rs;  // ... and this lane's own row
    dpp_ready(ld);
    d[J] = ld;
    a[J] = la;
    if constexpr (J < 15) {
      fmac_nbcast<J + 1>(d[J + 1], ld, ld);
      fmac_nbcast<J + 1>(a[J + 1], ld, la);
      const double pn = mov_nbcast<J + 1>(d[J + 1]);
      PanelUpdate<J, J + 2>::run(d, a, ld, la);
      PanelStep<J + 1>::run(d, a, pn);
    }
  }
};

// Panel kb (rows r0 = 16 kb .. 127, columns r0 .. r0 + 15), executed by waves 0 and 1.
//   in : S rows r0.. hold the trailing matrix (lower part valid);
//   out: global L gets the panel (zeros above the diagonal), S slot (kb, kb) gets W_d = L_d^-1 (lower, zeros above),
//        S slots (bi > kb, kb) get L[bi][kb].
// Lane l of wave w: replica row r0 + (l & 15) of the diagonal block (set d), and own row q = 64 w + l (set a):
// q < nrest: matrix row r0 + 16 + q;  q - nrest in [0, 16): row q - nrest of the identity;  else a zero row.
__device__ __forceinline__ void panel_factor(double* __restrict__ S, int r0, int w, double* __restrict__ L, int64_t ld,
                                             int64_t off, int* __restrict__ info) {
  const int lane = threadIdx.x & 63, lr = lane & 15;
  const int nrest = QN - 16 - r0, q = 64 * w + lane, e = q - nrest;
  const bool own = q < nrest, ident = e >= 0 && e < 16;
  const double* const rowd = S + (r0 + lr) * QS + r0;
  double* const rowa = S + (r0 + 16 + (own ? q : 0)) * QS + r0;
  double d[16], a[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const v2d x = *(const v2d*)(rowd + k);
    d[k] = x.x;
    d[k + 1] = x.y;
    const v2d y = own ? *(const v2d*)(rowa + k) : (v2d){0.0, 0.0};
    a[k] = (ident && e == k) ? 1.0 : y.x;
    a[k + 1] = (ident && e == k + 1) ? 1.0 : y.y;
  }
  PanelStep<0>::run(d, a, mov_nbcast<0>(d[0]));
  // diagonal of L (lane lr holds d[lr] = L[lr][lr]): a non-positive or NaN pivot leaves NaN from there on
  if (w == 0) {
    double dg = d[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) dg = (lr == k) ? d[k] : dg;
    const unsigned long long bad = __ballot(!(dg > 0.0)) & 0xffffull;
    if (bad != 0 && lane == 0) atomicCAS(info, 0, (int)(off + r0 + (__ffsll(bad) - 1)) + 1);
  }
  if (w == 0 && lane < 16) {  // the diagonal block itself: wave 0's first replica
    double* const gl = L + (off + r0 + lane) * ld + off + r0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      v2d x;
      x.x = (k <= lane) ? d[k] : 0.0;
      x.y = (k + 1 <= lane) ? d[k + 1] : 0.0;
      *(v2d*)(gl + k) = x;
    }
  }
  if (own) {
    double* const gl = L + (off + r0 + 16 + q) * ld + off + r0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      v2d x;
      x.x = a[k];
      x.y = a[k + 1];
      *(v2d*)(gl + k) = x;
      *(v2d*)(rowa + k) = x;
    }
  }
  if (ident) {  // W_d[k][e] = a[k]: zero above the diagonal by construction (the row started as e_e)
#pragma unroll
    for (int k = 0; k < 16; ++k) S[(r0 + k) * QS + r0 + e] = a[k];
  }
}

// ---- block products on one wave -------------------------------------------------------------------------------
// Byte offsets into S.  A 16 x 16 block (bi, bj) starts at blk(bi, bj); with lane = 16 lq + lr:
//   A operand / "transposed" B operand (B[k][col] = block[col][k]), k-group k4:  origin + lane_a + 32 k4
//   natural B operand (B[k][col] = block[k][col]),                 k-group k4:  origin + lane_o + 4224 k4
//   accumulator entry r <-> block[lq + 4 r][lr]:                                 origin + lane_o + 4224 r
constexpr int RB = QS * 8;  // bytes per LDS row
__device__ __forceinline__ constexpr int blk(int bi, int bj) { return (16 * bi * QS + 16 * bj) * 8; }
__device__ __forceinline__ double ldsd(const char* p) { return *(const double*)p; }

// Work items of the six "worker" waves during panel kb (kb >= 1), all independent of each other and of the panel
// (pb = kb - 1; see the schedule in leaf128_kernel):  out = keep * out - (A1 B1 [+ A2 B2])
struct WorkItem {
  int a1, b1, b1nat, o;       // block origins (bytes); b1nat: natural (1) or transposed (0) B operand
  int a2, b2, flags, pad;     // second product (natural B); flags: has2 | keep << 1 | valid << 2
};
// Workers are the waves 2 .. 7.  Measured alternatives (tools/ubench_leaf.hip; a panel alone takes ~4650 cycles):
//   waves 2..7, four items each (this):     panel 5000-5700 (waves 4, 5 share the panel waves' SIMDs), workers done first;
//   waves 2, 3, 6, 7 only, six items each:  panel 4400-4700, but the workers need ~6700 (one wave issues a float64
//                                           MFMA only every ~128 cycles) and the panel waves wait 1300-3300 for them;
//   twelve waves, workers 2,3,6,7,10,11:    panel 4700-5400, workers ~7000 (three waves per SIMD queue on its MFMA pipe);
//   all of a worker's loads up front:       workers ~5300, panel 5800-6300 (more LDS traffic against the panel waves).
constexpr int NWORK = 6, WSLOTS = 4;  // worker waves, items per worker and panel (<= 22 items per panel)
constexpr int SCRATCH_BLK = (0 * 16 * QS + 16 * 7) * 8;  // block (0, 7): the blocks above the block diagonal are never used

__device__ __forceinline__ WorkItem make_work_item(int kb, int idx) {
  const int pb = kb - 1;
  const int ns = (QB - 1 - kb) * (QB - kb) / 2, nt = (QB - kb) * pb;
  WorkItem it{};
  if (idx < ns) {  // trailing update (bi, bj), kb < bj <= bi, against panel pb:  S[bi][bj] -= L[bi][pb] L[bj][pb]^T
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= idx) ++r;
    const int bi = kb + 1 + r, bj = kb + 1 + idx - r * (r + 1) / 2;
    it.a1 = blk(bi, pb); it.b1 = blk(bj, pb); it.b1nat = 0; it.o = blk(bi, bj); it.flags = 2 | 4;
  } else if (idx < ns + nt) {  // row bi >= kb of T against source row pb
    const int bi = kb + (idx - ns) / pb, c = (idx - ns) % pb;
    it.a1 = blk(bi, pb); it.b1 = blk(pb, c); it.b1nat = 1; it.o = blk(bi, c);
    if (c < pb - 1) {  // T[bi][c] -= L[bi][pb] T[pb][c]
      it.flags = 2 | 4;
    } else {           // T[bi][pb-1] = -L[bi][pb-1] W_d[pb-1] - L[bi][pb] T[pb][pb-1]  (replaces the L block)
      it.a2 = blk(bi, pb - 1); it.b2 = blk(pb - 1, pb - 1); it.flags = 1 | 4;
    }
  } else if (idx == ns + nt) {
    // T~[kb][pb] = -L[kb][pb] W_d[pb], the one entry of row kb of T that no earlier step could prepare, into the
    // scratch block: row kb's slot (kb, pb) still holds L[kb][pb], which this panel's other items read
    it.a1 = blk(kb, pb); it.b1 = blk(pb, pb); it.b1nat = 1; it.o = SCRATCH_BLK; it.flags = 4;
  }
  return it;
}

// Schedule.  T = W is built from "contributions": slot (bi, c), c < bi, must end as
//     T[bi][c] = W_d[bi] ( -L[bi][c] W_d[c] - sum_{c < j < bi} L[bi][j] T[j][c] ),
// and it holds L[bi][c] until the last reader of that block is done.  Per panel kb (pb = kb - 1):
//   [P kb]  waves 0, 1: panel kb (block column kb only).
//           workers:    trailing updates of panel pb for block columns > kb;  contributions of source row pb to the
//                       rows bi >= kb (columns c < pb), the one for c = pb - 1 fused with the initial term
//                       -L[bi][pb-1] W_d[pb-1] that replaces the L block (nobody reads L[.][pb-1] any more);
//                       T~[kb][pb] for [C kb].
//   barrier
//   [C kb]  one item per wave: trailing update of panel kb for block column kb + 1 (all the next panel needs), and
//           row kb of T:  T[kb][c] = W_d[kb] T~[kb][c]  (T~[kb][pb] = -L[kb][pb] W_d[pb] comes from a worker, via a scratch block).
//   barrier
// so the critical path per panel is the panel itself, one 16 x 16 x 16 product and two barriers.
__global__ __launch_bounds__(512) void leaf128_kernel(const double* __restrict__ A, double* __restrict__ L,
                                                      double* __restrict__ W, int64_t ld, int64_t off,
                                                      int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double S[];  // [QN][QS], then the work-item table
  WorkItem* const items = (WorkItem*)(S + QN * QS);           // [QB][NWORK][WSLOTS]
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const S8 = (const char*)S;
  const int lane_a = (lr * QS + lq) * 8, lane_o = (lq * QS + lr) * 8;
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
  if (tid < QB * NWORK * WSLOTS) {
    const int kb = tid / (NWORK * WSLOTS), wk = (tid / WSLOTS) % NWORK, u = tid % WSLOTS;
    // waves 4 and 5 (wk 2, 3) share the panel waves' SIMDs: they come last in the hand-out, so that they are the
    // ones left without work when a panel has few items
    const int rank = wk < 2 ? wk : (wk < 4 ? wk + 2 : wk - 2);
    items[tid] = kb >= 1 ? make_work_item(kb, rank + NWORK * u) : WorkItem{};
  }
  __syncthreads();
  TGP_LEAF_TICK(0);
#pragma unroll 1
  for (int kb = 0; kb < QB; ++kb) {
    const int r0 = 16 * kb;
    if (w < 2) {
      __builtin_amdgcn_s_setprio(3);  // the chain: ahead of the workers sharing its SIMD
      panel_factor(S, r0, w, L, ld, off, info);
      __builtin_amdgcn_s_setprio(0);
    } else if (kb > 0) {
      const v4i* const tab = (const v4i*)(items + (kb * NWORK + (w - 2)) * WSLOTS);
      v4d acc[WSLOTS];
      double old[WSLOTS][4];
      int oaddr[WSLOTS], flags[WSLOTS];
#pragma unroll
      for (int u = 0; u < WSLOTS; ++u) {
        const v4i e0 = tab[2 * u], e1 = tab[2 * u + 1];  // per-lane copies of wave-uniform values
        flags[u] = 0;
        if (!(__builtin_amdgcn_readfirstlane(e1.z) & 4)) continue;  // no item in this slot (wave-uniform)
        const char* const pa = S8 + lane_a + e0.x;
        const char* const pb1 = S8 + (e0.z ? lane_o : lane_a) + e0.y;
        const int step = e0.z ? 4 * RB : 32;
        double av[4], bv[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          av[k4] = ldsd(pa + 32 * k4);
          bv[k4] = ldsd(pb1 + k4 * step);
        }
        oaddr[u] = lane_o + e0.w;
        flags[u] = e1.z;
#pragma unroll
        for (int r = 0; r < 4; ++r) old[u][r] = ldsd(S8 + oaddr[u] + 4 * RB * r);
        acc[u] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) acc[u] = mfma_f64(av[k4], bv[k4], acc[u]);
        if (__builtin_amdgcn_readfirstlane(e1.z) & 1) {  // wave-uniform: the fused initial term
          const char* const pa2 = S8 + lane_a + e1.x;
          const char* const pb2 = S8 + lane_o + e1.y;
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) acc[u] = mfma_f64(ldsd(pa2 + 32 * k4), ldsd(pb2 + 4 * RB * k4), acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < WSLOTS; ++u) {
        const double keep = (flags[u] & 2) ? 1.0 : 0.0;
        if (flags[u] & 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) *(double*)((char*)S + oaddr[u] + 4 * RB * r) = fma(keep, old[u][r], -acc[u][r]);
        }
      }
    }
    TGP_LEAF_TICK(1 + 4 * kb);
    __syncthreads();
    TGP_LEAF_TICK(2 + 4 * kb);
    {
      const int pb = kb - 1, ncol = QB - 1 - kb;  // column items: waves [0, ncol); row kb of T: waves [ncol, ncol + kb)
      if (w < ncol) {  // S[bi][kb+1] -= L[bi][kb] L[kb+1][kb]^T
        const int bi = kb + 1 + w;
        const char* const pa = S8 + lane_a + blk(bi, kb);
        const char* const pb1 = S8 + lane_a + blk(kb + 1, kb);
        char* const po = (char*)S + lane_o + blk(bi, kb + 1);
        double av[4], bv[4], old[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          av[k4] = ldsd(pa + 32 * k4);
          bv[k4] = ldsd(pb1 + 32 * k4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = ldsd(po + 4 * RB * r);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) acc = mfma_f64(av[k4], bv[k4], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) *(double*)(po + 4 * RB * r) = old[r] - acc[r];
      } else if (w < ncol + kb) {
        const int c = w - ncol;
        const char* const pwd = S8 + lane_a + blk(kb, kb);  // W_d[kb] as the A operand
        char* const po = (char*)S + lane_o + blk(kb, c);
        double wd[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) wd[k4] = ldsd(pwd + 32 * k4);
        // T~[kb][c]: complete in its slot for c < pb; the last one, -L[kb][pb] W_d[pb], was left in the scratch block by a
        // worker during the panel (the accumulator layout IS the natural B operand layout: entry r <-> k-group r)
        const char* const px = (c < pb) ? (const char*)po : S8 + lane_o + SCRATCH_BLK;
        v4d x;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = ldsd(px + 4 * RB * r);
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) acc = mfma_f64(wd[k4], x[k4], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) *(double*)(po + 4 * RB * r) = acc[r];
      }
    }
    TGP_LEAF_TICK(3 + 4 * kb);
    __syncthreads();
    TGP_LEAF_TICK(4 + 4 * kb);
  }
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
  constexpr size_t shmem = (size_t)QN * QS * sizeof(double) + QB * NWORK * WSLOTS * sizeof(WorkItem);
  // > 64 KiB of dynamic LDS is opt-in, per function and per device: set on every launch
  (void)hipFuncSetAttribute((const void*)leaf128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(leaf128_kernel, dim3(1), dim3(512), shmem, s, A, L, W, ld, off, info);
}

}  // namespace tgp
