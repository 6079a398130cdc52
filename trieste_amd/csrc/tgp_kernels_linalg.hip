// `update` path kernels for gfx950: input scaling, K assembly, 64x64 Cholesky+inverse leaf,
// f64-MFMA GEMM (the trailing updates of the recursive Cholesky / triangular inverse),
// masked transpose, triangular mat-vec.
//
// Replaces (reference call sites): self.model.kernel(x) and tf.linalg.cholesky(K + s) inside
// gpflow GPRPosterior._precompute reached from trieste/models/gpflow/models.py:171-186 ->
// interface.py:108-112.  
#include <cstdlib>
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

namespace tgp {

// ---------------------------------------------------------------------------------------------
__global__ void scale_inputs_kernel(const double* __restrict__ X, const double* __restrict__ ls,
                                    double* __restrict__ Xs, int64_t N, int64_t Npad, int d, int dp) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Npad * dp) return;
  const int64_t i = t / dp;
  const int c = (int)(t % dp);
  Xs[t] = (i < N && c < d) ? X[i * d + c] / ls[c] : 0.0;  // gpflow Stationary.scale: X / lengthscales
}

void launch_scale_inputs(hipStream_t s, const double* X, const double* ls, double* Xs, int64_t N,
                         int64_t Npad, int d, int dp) {
  const int64_t n = Npad * dp;
  hipLaunchKernelGGL(scale_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, ls,
                     Xs, N, Npad, d, dp);
}

// Batched trial evaluations (tgp_nlml_trial_batch): member b = blockIdx.y has its own lengthscales / mean (hyp + b
// hyp_stride: ls [dp] at 0, variance at 32, noise at 33, mean at 34) -- its scaled inputs and centred targets in ONE
// launch for all members (the same arithmetic as scale_inputs_kernel / center_kernel).
__global__ void batch_prep_kernel(const double* __restrict__ X, const double* __restrict__ Y, const double* __restrict__ hyp,
                                  int64_t hyp_stride, double* __restrict__ Xs, double* __restrict__ err, int64_t N,
                                  int64_t Npad, int d, int dp) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double* h = hyp + (int64_t)blockIdx.y * hyp_stride;
  if (t < Npad * dp) {
    const int64_t i = t / dp;
    const int c = (int)(t % dp);
    Xs[(int64_t)blockIdx.y * Npad * dp + t] = (i < N && c < d) ? X[i * d + c] / h[c] : 0.0;
  }
  if (t < Npad) err[(int64_t)blockIdx.y * Npad + t] = (t < N) ? Y[t] - h[34] : 0.0;
}
void launch_batch_prep(hipStream_t s, const double* X, const double* Y, const double* hyp, int64_t hyp_stride, int B,
                       double* Xs, double* err, int64_t N, int64_t Npad, int d, int dp) {
  const int64_t n = Npad * dp;
  hipLaunchKernelGGL(batch_prep_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, s, X, Y, hyp, hyp_stride,
                     Xs, err, N, Npad, d, dp);
}

// K[i][j] = k(x_i, x_j) + noise * (i == j) on the N x N part; identity on the padding so that the
// padded matrix stays SPD and factorises to blockdiag(L, I).  Only the 64 x 64 tiles on and below the
// diagonal are written: nothing downstream reads the upper triangle (the leaf mirrors its diagonal tiles
// inside LDS, the node products read A21 / the lower tiles of A22).  One workgroup per tile: lane -> column
// (coalesced stores, x_j in registers), wave -> 16 rows (wave-uniform: x_i through scalar loads); the same
// branch-free kernel_from_r2<KIND> as the sweep, so K and K* are the same function bit for bit.
// Batched form (hyp != nullptr; tgp_nlml_trial_batch): member blockIdx.z reads its variance / noise from hyp (+ z
// hyp_stride: slots 32, 33), its scaled inputs at Xs + z Npad DP, and writes A + z a_stride.
template <int KIND, int DP>
__global__ __launch_bounds__(256) void assemble_K_kernel(const double* __restrict__ Xs, double* __restrict__ A,
                                                         int64_t N, int64_t Npad, double variance, double noise,
                                                         int64_t row0, const double* __restrict__ hyp = nullptr,
                                                         int64_t hyp_stride = 0, int64_t a_stride = 0) {
  const int64_t tj = blockIdx.x, ti = row0 / 64 + blockIdx.y;
  if (tj > ti) return;
  if (hyp) {
    const int64_t z = blockIdx.z;
    variance = hyp[z * hyp_stride + 32];
    noise = hyp[z * hyp_stride + 33];
    Xs += z * Npad * DP;
    A += z * a_stride;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t j = tj * 64 + lane;
  double xj[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) xj[c] = Xs[j * DP + c];
  const cptr xs = as_const(Xs);
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int64_t i = ti * 64 + w * 16 + r;
    double v;
    if (i < N && j < N) {
      double r2 = 0.0;
#pragma unroll
      for (int c = 0; c < DP; ++c) {
        const double t = xs[i * DP + c] - xj[c];
        r2 = fma(t, t, r2);
      }
      v = kernel_from_r2<KIND>(r2, variance);
      if (i == j) v += noise;
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    A[i * Npad + j] = v;
  }
}

template <int KIND>
static void launch_assemble_K_dp(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp,
                                 double variance, double noise, int64_t row0, const double* hyp, int64_t hyp_stride,
                                 int64_t a_stride, int B) {
  // A is addressed with GLOBAL row indices: for row0 > 0 the caller passes (scratch - row0 * Npad)
  dim3 g((unsigned)(Npad / 64), (unsigned)((Npad - row0) / 64), (unsigned)B), b(256);
#define TGP_ASM_K(DPV) \
  hipLaunchKernelGGL((assemble_K_kernel<KIND, DPV>), g, b, 0, s, Xs, A, N, Npad, variance, noise, row0, hyp, hyp_stride, a_stride)
  switch (dp) {
    case 2: TGP_ASM_K(2); break;
    case 4: TGP_ASM_K(4); break;
    case 6: TGP_ASM_K(6); break;
    case 8: TGP_ASM_K(8); break;
    case 16: TGP_ASM_K(16); break;
    default: TGP_ASM_K(32); break;
  }
#undef TGP_ASM_K
}

static void launch_assemble_K_any(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp, int kind,
                                  double variance, double noise, int64_t row0, const double* hyp, int64_t hyp_stride,
                                  int64_t a_stride, int B) {
  switch (kind) {
    case KIND_RBF: launch_assemble_K_dp<KIND_RBF>(s, Xs, A, N, Npad, dp, variance, noise, row0, hyp, hyp_stride, a_stride, B); break;
    case KIND_M12: launch_assemble_K_dp<KIND_M12>(s, Xs, A, N, Npad, dp, variance, noise, row0, hyp, hyp_stride, a_stride, B); break;
    case KIND_M32: launch_assemble_K_dp<KIND_M32>(s, Xs, A, N, Npad, dp, variance, noise, row0, hyp, hyp_stride, a_stride, B); break;
    default: launch_assemble_K_dp<KIND_M52>(s, Xs, A, N, Npad, dp, variance, noise, row0, hyp, hyp_stride, a_stride, B); break;
  }
}

void launch_assemble_K(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp,
                       int kind, double variance, double noise, int64_t row0) {
  launch_assemble_K_any(s, Xs, A, N, Npad, dp, kind, variance, noise, row0, nullptr, 0, 0, 1);
}
// B members in one launch: member b's inputs at Xs + b Npad dp, its matrix at A + b a_stride, its (variance, noise) at
// hyp [b hyp_stride + 32 / 33]
void launch_assemble_K_batch(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp, int kind,
                             const double* hyp, int64_t hyp_stride, int64_t a_stride, int B) {
  launch_assemble_K_any(s, Xs, A, N, Npad, dp, kind, 0.0, 0.0, 0, hyp, hyp_stride, a_stride, B);
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// 64 x 64 leaf (Cholesky factor L of a diagonal block AND W = L^-1, one workgroup of 256 threads; info: 0 = ok, else
// 1 + global index of the first non-positive pivot).  Used for the odd 64-block of a node whose size is not a multiple
// of 128 (rank-k appends, small joint covariances); everything else goes through the 128-leaf of tgp_kernels_leaf.hip.
// 16-wide panels, the 64 x 64 block resident in LDS.
//   per panel:  (1) wave 0 factors the 16 x 16 diagonal block AND inverts it with a register-resident rank-1
//                   scheme (64 lanes, 4 + 4 entries each; column j of L and row j of the running inverse travel
//                   through a double-buffered LDS strip) -- a single wave, so its 16 pivot steps need no workgroup
//                   barrier;
//               (2) panel  L_p = A_p W_d^T  (VALU, <= 48 x 16 outputs);
//               (3) trailing update  A_22 -= L_p L_p^T  by MFMA (<= 6 tiles of 16 x 16, k = 16);
//   then the blocked triangular inverse  W[bi][bk] = -W_d[bi] sum_j L[bi][j] W[j][bk]  by MFMA (the
//   accumulator layout of the first product is the B-operand layout of the second: no LDS round trip).
constexpr int LS = LEAF + 4;  // LDS row stride (16-byte aligned rows)
__global__ __launch_bounds__(256) void leaf_blocked_kernel(const double* __restrict__ A, double* __restrict__ L,
                                                           double* __restrict__ W, int64_t ld, int64_t off,
                                                           int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double S[LEAF][LS];
  __shared__ __attribute__((aligned(16))) double T[LEAF][LS];
  __shared__ __attribute__((aligned(16))) double colb[2][16];
  __shared__ __attribute__((aligned(16))) double rowb[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int e = tid; e < LEAF * LEAF; e += 256) {  // rows of the lower triangle: coalesced
    const int i = e >> 6, j = e & 63;
    S[i][j] = (j <= i) ? A[(off + i) * ld + off + j] : 0.0;
    T[i][j] = 0.0;
  }
  __syncthreads();
  {  // only the diagonal 16 x 16 tiles are read as full symmetric tiles: mirror them inside LDS
    const int bt = tid >> 6, i = (tid >> 2) & 15, p = tid & 3;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = p + 4 * m;
      if (k > i) S[16 * bt + i][16 * bt + k] = S[16 * bt + k][16 * bt + i];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    const int r0 = 16 * kb, r1 = r0 + 16;
    if (w == 0) {  // (1) diagonal block: factor + inverse, one wave, no workgroup barrier
      const int i = lane >> 2, p = lane & 3;
      double a[4], t[4], myrs = 1.0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a[m] = S[r0 + i][r0 + p + 4 * m];
        t[m] = (p + 4 * m == i) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jm = j >> 2, jp = j & 3, bsel = j & 1;
        if (p == jp) colb[bsel][i] = a[jm];
        if (i == j) {
#pragma unroll
          for (int m = 0; m <= jm; ++m) rowb[bsel][p + 4 * m] = t[m];
        }
        __builtin_amdgcn_wave_barrier();  // LDS executes a wave's accesses in order: the reads below see the writes
        double dj = colb[bsel][j];
        if (!(dj > 0.0)) {
          if (lane == 0) atomicCAS(info, 0, (int)(off + r0 + j) + 1);
          dj = 1.0;
        }
        double sd, rs;
        sqrt_and_rsqrt_short(dj, sd, rs);
        const double lij = colb[bsel][i];
        const double f = lij * (rs * rs);
#pragma unroll
        for (int m = jm; m < 4; ++m) {
          const double upd = fma(-f, colb[bsel][p + 4 * m], a[m]);
          a[m] = (m > jm || p > jp) ? upd : a[m];
        }
        if (p == jp) a[jm] = (i == j) ? sd : lij * rs;
        const double ft = (i > j) ? f : 0.0;
        myrs = (i == j) ? rs : myrs;
#pragma unroll
        for (int m = 0; m <= jm; ++m) t[m] = fma(-ft, rowb[bsel][p + 4 * m], t[m]);
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = p + 4 * m;
        S[r0 + i][r0 + k] = (k <= i) ? a[m] : 0.0;
        T[r0 + i][r0 + k] = (k <= i) ? t[m] * myrs : 0.0;
      }
    }
    __syncthreads();
    if (r1 < LEAF) {
      {  // (2) panel: L_p[row][c] = sum_{k <= c} A_p[row][k] W_d[c][k]
        const int row = tid >> 2, part = tid & 3;
        double out[4] = {0.0, 0.0, 0.0, 0.0};
        if (row >= r1) {
          double in[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) in[k] = S[row][r0 + k];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int c = part * 4 + cc;
#pragma unroll
            for (int k = 0; k < 16; ++k) out[cc] = fma(in[k], (k <= c) ? T[r0 + c][r0 + k] : 0.0, out[cc]);
          }
        }
        __syncthreads();
        if (row >= r1) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) S[row][r0 + part * 4 + cc] = out[cc];
        }
        __syncthreads();
      }
      // (3) trailing update on the lower 16 x 16 tiles (bi >= bj > kb), MFMA, k = 16
      int tix = 0;
      for (int bi = kb + 1; bi < 4; ++bi)
        for (int bj = kb + 1; bj <= bi; ++bj, ++tix) {
          if ((tix & 3) != w) continue;
          v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const double av = S[16 * bi + (lane & 15)][r0 + 4 * k4 + (lane >> 4)];
            const double bv = S[16 * bj + (lane & 15)][r0 + 4 * k4 + (lane >> 4)];
            acc = mfma_f64(av, bv, acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) S[16 * bi + (lane >> 4) + 4 * r][16 * bj + (lane & 15)] -= acc[r];
        }
      __syncthreads();
    }
  }
  // blocked triangular inverse: block row bi, block column bk = wave index
#pragma unroll 1
  for (int bi = 1; bi < 4; ++bi) {
    if (w < bi) {
      const int bk = w;
      v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
      for (int j = bk; j < bi; ++j) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const double av = S[16 * bi + (lane & 15)][16 * j + 4 * k4 + (lane >> 4)];   // L[bi][j]
          const double bv = T[16 * j + 4 * k4 + (lane >> 4)][16 * bk + (lane & 15)];   // W[j][bk]
          acc = mfma_f64(av, bv, acc);
        }
      }
      v4d out = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const double av = T[16 * bi + (lane & 15)][16 * bi + 4 * k4 + (lane >> 4)];     // W_d[bi]
        out = mfma_f64(av, acc[k4], out);  // acc[k4] holds rows 4 k4 .. 4 k4 + 3: the B operand of this k group
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) T[16 * bi + (lane >> 4) + 4 * r][16 * bk + (lane & 15)] = -out[r];
    }
    __syncthreads();
  }
  for (int e = tid; e < LEAF * LEAF; e += 256) {
    const int i = e >> 6, j = e & 63;
    L[(off + i) * ld + off + j] = (j <= i) ? S[i][j] : 0.0;
    W[(off + i) * ld + off + j] = (j <= i) ? T[i][j] : 0.0;
  }
}

void launch_leaf(hipStream_t s, const double* A, double* L, double* W, int64_t ld, int64_t off,
                 int* info) {
  hipLaunchKernelGGL(leaf_blocked_kernel, dim3(1), dim3(256), 0, s, A, L, W, ld, off, info);
}

// ---------------------------------------------------------------------------------------------
// f64 MFMA GEMMs of the factor recursion (v_mfma_f64_16x16x4_f64).  m, n multiples of 64; k multiple of 32.
// Operand tiles are staged in LDS k-major ([k][m] / [k][n]) with a +16-double row pad so that the 4 k-rows
// a fragment read touches fall on disjoint bank halves.
constexpr int GB = 64, GLD = GB + 16;


// Workgroup -> output tile.  The launch is a 1-D grid over the LIVE tiles only, ordered so that tiles
// with the longest k range start first and equal-length tiles are adjacent (the dispatcher hands
// workgroups out in order: a short..long pattern repeating every tile row, or a grid whose dead half
// exits at once, left the makespan at the unpruned launch's -- measured 2110 vs 2126 us at 4096^3):
//   lower_only: the nt (nt + 1) / 2 tiles with tn <= tm, row by row (all the same length);
//   tri 1 (k < (tn+1) T): column-major, last column first;   tri 2 (k >= tn T): column-major, first column first;
//   tri 3 (k < (tm+1) T): row-major, last row first;          tri 5 (k >= tm T): row-major, first row first;
//   tri 4 (k >= max(tm, tn) T) and unpruned: row-major with the column rotated by the row.
__device__ __forceinline__ void tile_of_block(int lower_only, int tri, int nt_m, int nt_n, int& tm, int& tn,
                                              int id = -1) {
  if (id < 0) id = blockIdx.x;
  if (lower_only) {
    int r = (int)((sqrt(8.0 * id + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= id) ++r;
    while (r * (r + 1) / 2 > id) --r;
    tm = r;
    tn = id - r * (r + 1) / 2;
  } else if (tri == 1 || tri == 2) {
    const int c = id / nt_m;
    tm = id % nt_m;
    tn = (tri == 1) ? nt_n - 1 - c : c;
  } else if (tri == 3 || tri == 5) {
    const int r = id / nt_n;
    tn = id % nt_n;
    tm = (tri == 3) ? nt_m - 1 - r : r;
  } else {
    tm = id / nt_n;
    tn = (id % nt_n + tm) % nt_n;
  }
}

// `tri` prunes the k range per output tile when one operand is lower triangular (the factor
// recursion of tgp_api.hip):  1: B^T with B lower (k < (tn+1) 64);  2: B lower, not transposed
// (k >= tn 64);  3: A lower (k < (tm+1) 64);  4: A upper and B lower (k >= max(tm, tn) 64);
// 5: A upper (k >= tm 64).  Global loads of step k0+16 are in flight while the
// MFMAs of step k0 run (register prefetch), so a step costs max(load, math) instead of the sum.
// 64 x 64 tile on EIGHT waves (2 x 4, wave tile 32 x 16), 32-deep k steps, register prefetch of the next
// step's global loads.  Eight rather than four waves: with at most one workgroup per CU (every node of the
// recursion below 2048) a 4-wave workgroup leaves one MFMA-issuing wave per SIMD, and one wave issues an
// f64 MFMA only every 128 cycles -- half the pipe rate; measured, the 8-wave form wins at every size.
//
// Transposing stores into LDS: the k rows lk, lk+4, ... of the lanes that share a tile row are 4 LDS rows
// apart = the same bank with the +16 pad that keeps the fragment READS conflict-free (an 8-way conflict on
// every store; it capped every small product at 1.7 us per 32-deep step whatever the wave count).  Element
// (k, col) therefore lives at column (col + 8 (k >> 2)) & 63: a rotation that is uniform inside a 4-row k
// group, so the reads stay conflict-free.
//
// gridDim.y > 1: k-split -- slice blockIdx.y of the (pruned) k range, partial result to C + slice * m * ldc.
constexpr int GKT = 32;
// depth of a k-split slice: the k range in nz equal parts, rounded up to whole 32-deep steps
__host__ __device__ __forceinline__ int ksplit_depth(int k, int nz) { return ((k / GKT + nz - 1) / nz) * GKT; }
template <bool TB>
__device__ __forceinline__ void gemm8_body(int m, int n, int k, double alpha, const double* __restrict__ A,
                                           int64_t lda, const double* __restrict__ B, int64_t ldb, double beta,
                                           double* __restrict__ C, int64_t ldc, int lower_only, int tri,
                                           int block_id, double (*As)[GLD], double (*Bs)[GLD]) {
  int tm, tn;
  tile_of_block(lower_only, tri, m / GB, n / GB, tm, tn, block_id);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 2, wn = w & 3;
  v4d acc[2];
  acc[0] = (v4d){0.0, 0.0, 0.0, 0.0};
  acc[1] = (v4d){0.0, 0.0, 0.0, 0.0};
  int klo = 0, khi = k;
  if (tri == 1) khi = min(k, (tn + 1) * GB);
  else if (tri == 2) klo = min(k, tn * GB);
  else if (tri == 3) khi = min(k, (tm + 1) * GB);
  else if (tri == 4) klo = min(k, max(tm, tn) * GB);
  else if (tri == 5) klo = min(k, tm * GB);
  if (gridDim.y > 1) {
    // slices of FIXED depth over the absolute k range (round 6, last session: a slice of the tile's own pruned range before -- every
    // tile wrote gridDim.y partial tiles however short its range, 67 MB of partials for a 4096 x 128 product at 16 slices, and the
    // workgroups of the last tile rows walked 16 times the depth of the first); a slice that misses the tile's range writes nothing,
    // ksplit_reduce_kernel sums the live ones (ksplit_live below is the one rule both use)
    const int per = ksplit_depth(k, (int)gridDim.y);
    const int lo2 = (int)blockIdx.y * per, hi2 = lo2 + per;
    if (hi2 <= klo || lo2 >= khi) return;
    klo = max(klo, lo2);
    khi = min(khi, hi2);
    C += (int64_t)blockIdx.y * m * ldc;
  }
  const double* Ab = A + (int64_t)tm * GB * lda;
  const double* Bb = TB ? B + (int64_t)tn * GB * ldb : B + (int64_t)tn * GB;
  const int lr = tid >> 3, lk = (tid & 7) * 4;   // [row][k..k+3] loader: 64 rows x 32 k
  const int br = tid >> 4, bc = (tid & 15) * 4;  // [k][n..n+3] loader: 32 k x 64 n
  v2d a0, a1, b0, b1;
  auto fetch = [&](int k0) {
    const double* sa = Ab + (int64_t)lr * lda + k0 + lk;
    a0 = *(const v2d*)sa;
    a1 = *(const v2d*)(sa + 2);
    const double* sb = TB ? Bb + (int64_t)lr * ldb + k0 + lk : Bb + (int64_t)(k0 + br) * ldb + bc;
    b0 = *(const v2d*)sb;
    b1 = *(const v2d*)(sb + 2);
  };
  if (klo < khi) fetch(klo);
  for (int k0 = klo; k0 < khi; k0 += GKT) {
    const int sc = (lr + 2 * lk) & 63;  // the column rotation by 8 (k >> 2)
    As[lk + 0][sc] = a0.x; As[lk + 1][sc] = a0.y; As[lk + 2][sc] = a1.x; As[lk + 3][sc] = a1.y;
    if (TB) {
      Bs[lk + 0][sc] = b0.x; Bs[lk + 1][sc] = b0.y; Bs[lk + 2][sc] = b1.x; Bs[lk + 3][sc] = b1.y;
    } else {
      *(v2d*)&Bs[br][bc] = b0;
      *(v2d*)&Bs[br][bc + 2] = b1;
    }
    __syncthreads();
    if (k0 + GKT < khi) fetch(k0 + GKT);
#pragma unroll
    for (int k4 = 0; k4 < GKT / 4; ++k4) {
      const int kr = k4 * 4 + (lane >> 4);
      const double bv = Bs[kr][TB ? (wn * 16 + (lane & 15) + 8 * k4) & 63 : wn * 16 + (lane & 15)];
#pragma unroll
      for (int f = 0; f < 2; ++f)
        acc[f] = mfma_f64(As[kr][(wm * 32 + f * 16 + (lane & 15) + 8 * k4) & 63], bv, acc[f]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = (int64_t)tm * GB + wm * 32 + f * 16 + (lane >> 4) + 4 * r;
      const int64_t col = (int64_t)tn * GB + wn * 16 + (lane & 15);
      double* dst = C + row * ldc + col;
      const double v = alpha * acc[f][r];
      *dst = (beta == 0.0) ? v : fma(beta, *dst, v);
    }
}

template <bool TB>
__global__ __launch_bounds__(512) void gemm_kernel8(int m, int n, int k, double alpha,
                                                    const double* __restrict__ A, int64_t lda,
                                                    const double* __restrict__ B, int64_t ldb,
                                                    double beta, double* __restrict__ C, int64_t ldc,
                                                    int lower_only, int tri) {
  __shared__ double As[GKT][GLD];
  __shared__ double Bs[GKT][GLD];
  gemm8_body<TB>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower_only, tri, (int)blockIdx.x, As, Bs);
}

// Small-grid variant: 32 x 32 tile on eight waves = four 16 x 16 fragments x two halves of every 32-deep k step
// (reduced through LDS at the end).  Below ~256 live 64 x 64 tiles the kernel above leaves most CUs idle while
// each workgroup walks its whole k range alone (64 x 64 x k on one CU: 3.4 us per 128 of k at the f64 MFMA rate,
// the 1024-node's longest tile 27 us); four times as many workgroups with a quarter of the work each, and the
// triangular pruning at 32 instead of 64, shorten exactly that.  Same operand layout and rotation as above.
constexpr int SB = 32, SLD = SB + 16;
template <bool TB>
__device__ __forceinline__ void gemm_small_body(int m, int n, int k, double alpha, const double* __restrict__ A,
                                                int64_t lda, const double* __restrict__ B, int64_t ldb, double beta,
                                                double* __restrict__ C, int64_t ldc, int lower_only, int tri,
                                                int block_id, double (*As)[SLD], double (*Bs)[SLD], double* red) {
  int tm, tn;
  tile_of_block(lower_only, tri, m / SB, n / SB, tm, tn, block_id);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fm = (w >> 1) & 1, fn = w & 1, kh = w >> 2;
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  int klo = 0, khi = k;
  if (tri == 1) khi = min(k, (tn + 1) * SB);
  else if (tri == 2) klo = min(k, tn * SB);
  else if (tri == 3) khi = min(k, (tm + 1) * SB);
  else if (tri == 4) klo = min(k, max(tm, tn) * SB);
  else if (tri == 5) klo = min(k, tm * SB);
  const double* Ab = A + (int64_t)tm * SB * lda;
  const double* Bb = TB ? B + (int64_t)tn * SB * ldb : B + (int64_t)tn * SB;
  const int lr = tid >> 4, lk = (tid & 15) * 2;  // [row][k, k+1] loader (32 x 32), also [k][n, n+1] for B when !TB
  v2d a0, b0;
  auto fetch = [&](int k0) {
    a0 = *(const v2d*)(Ab + (int64_t)lr * lda + k0 + lk);
    b0 = *(const v2d*)(TB ? Bb + (int64_t)lr * ldb + k0 + lk : Bb + (int64_t)(k0 + lr) * ldb + lk);
  };
  if (klo < khi) fetch(klo);
  for (int k0 = klo; k0 < khi; k0 += SB) {
    const int sc = (lr + 8 * (lk >> 2)) & (SB - 1);  // element (k, col) lives at column (col + 8 (k >> 2)) & 31
    As[lk][sc] = a0.x;
    As[lk + 1][sc] = a0.y;
    if (TB) {
      Bs[lk][sc] = b0.x;
      Bs[lk + 1][sc] = b0.y;
    } else {
      *(v2d*)&Bs[lr][lk] = b0;
    }
    __syncthreads();
    if (k0 + SB < khi) fetch(k0 + SB);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k4 = kh * 4 + q, kr = k4 * 4 + (lane >> 4);
      const double bv = Bs[kr][TB ? (fn * 16 + (lane & 15) + 8 * k4) & (SB - 1) : fn * 16 + (lane & 15)];
      acc = mfma_f64(As[kr][(fm * 16 + (lane & 15) + 8 * k4) & (SB - 1)], bv, acc);
    }
    __syncthreads();
  }
  if (kh == 1) *(v4d*)&red[((w & 3) * 64 + lane) * 4] = acc;
  __syncthreads();
  if (kh == 0) {
    const v4d o = *(const v4d*)&red[((w & 3) * 64 + lane) * 4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = (int64_t)tm * SB + fm * 16 + (lane >> 4) + 4 * r;
      const int64_t col = (int64_t)tn * SB + fn * 16 + (lane & 15);
      double* dst = C + row * ldc + col;
      const double v = alpha * (acc[r] + o[r]);
      *dst = (beta == 0.0) ? v : fma(beta, *dst, v);
    }
  }
}

template <bool TB>
__global__ __launch_bounds__(512) void gemm_small_kernel(int m, int n, int k, double alpha,
                                                         const double* __restrict__ A, int64_t lda,
                                                         const double* __restrict__ B, int64_t ldb,
                                                         double beta, double* __restrict__ C, int64_t ldc,
                                                         int lower_only, int tri) {
  __shared__ double As[SB][SLD];
  __shared__ double Bs[SB][SLD];
  __shared__ __attribute__((aligned(32))) double red[4 * 64 * 4];
  gemm_small_body<TB>(m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lower_only, tri, (int)blockIdx.x, As, Bs, red);
}

// how many live 64 x 64 tiles a product may have and still take the 32 x 32 form
static int small_tile_limit() {
  static const int v = getenv("TGP_GEMM_SMALL") ? atoi(getenv("TGP_GEMM_SMALL")) : 256;  // tuning aid
  return v;
}

// Two INDEPENDENT products of a recursion node in one launch: A22 -= L21 L21^T (lower tiles) and
// T = L21 W11 both need only L21; as two dependent launches the second waited for the first although each is
// a latency-bound chain on a handful of CUs.  Workgroups [0, nt1) run the first problem, the rest the second.
struct Gemm8Args {
  int m, n, k;
  double alpha;
  const double* A;
  int64_t lda;
  const double* B;
  int64_t ldb;
  double beta;
  double* C;
  int64_t ldc;
  int lower_only, tri;
};
__global__ __launch_bounds__(512) void gemm_dual_kernel(Gemm8Args p1 /* A B^T */, int nt1, Gemm8Args p2 /* A B */) {
  __shared__ double As[GKT][GLD];
  __shared__ double Bs[GKT][GLD];
  if ((int)blockIdx.x < nt1)
    gemm8_body<true>(p1.m, p1.n, p1.k, p1.alpha, p1.A, p1.lda, p1.B, p1.ldb, p1.beta, p1.C, p1.ldc, p1.lower_only,
                     p1.tri, (int)blockIdx.x, As, Bs);
  else
    gemm8_body<false>(p2.m, p2.n, p2.k, p2.alpha, p2.A, p2.lda, p2.B, p2.ldb, p2.beta, p2.C, p2.ldc, p2.lower_only,
                      p2.tri, (int)blockIdx.x - nt1, As, Bs);
}

__global__ __launch_bounds__(512) void gemm_dual_small_kernel(Gemm8Args p1 /* A B^T */, int nt1, Gemm8Args p2 /* A B */) {
  __shared__ double As[SB][SLD];
  __shared__ double Bs[SB][SLD];
  __shared__ __attribute__((aligned(32))) double red[4 * 64 * 4];
  if ((int)blockIdx.x < nt1)
    gemm_small_body<true>(p1.m, p1.n, p1.k, p1.alpha, p1.A, p1.lda, p1.B, p1.ldb, p1.beta, p1.C, p1.ldc, p1.lower_only,
                          p1.tri, (int)blockIdx.x, As, Bs, red);
  else
    gemm_small_body<false>(p2.m, p2.n, p2.k, p2.alpha, p2.A, p2.lda, p2.B, p2.ldb, p2.beta, p2.C, p2.ldc, p2.lower_only,
                           p2.tri, (int)blockIdx.x - nt1, As, Bs, red);
}

// node step 2 + 3:  A22 -= L21 L21^T (lower)  and  T = L21 W11 (W11 lower triangular: tri 2)
void launch_node_pair(hipStream_t s, int s2, int s1, const double* L21, double* A22, const double* W11, double* T,
                      int64_t ld) {
  Gemm8Args p1{s2, s2, s1, -1.0, L21, ld, L21, ld, 1.0, A22, ld, 1, 0};
  Gemm8Args p2{s2, s1, s1, 1.0, L21, ld, W11, ld, 0.0, T, ld, 0, 2};
  const int t2 = s2 / GB, t1 = s1 / GB;
  const int nt1 = t2 * (t2 + 1) / 2, nt2 = t2 * t1;
  if (nt1 + nt2 <= small_tile_limit()) {
    const int u2 = s2 / SB, u1 = s1 / SB;
    const int ns1 = u2 * (u2 + 1) / 2, ns2 = u2 * u1;
    hipLaunchKernelGGL(gemm_dual_small_kernel, dim3((unsigned)(ns1 + ns2)), dim3(512), 0, s, p1, ns1, p2);
    return;
  }
  hipLaunchKernelGGL(gemm_dual_kernel, dim3((unsigned)(nt1 + nt2)), dim3(512), 0, s, p1, nt1, p2);
}

// Large-grid variant: 128 x 128 tile, 8 waves (2 x 4), wave tile 64 x 32 (the sweep kernel's wave tile:
// 6 LDS fragment reads per 8 MFMAs), 16-deep k steps, double-buffered LDS (one barrier per step), register
// prefetch one step ahead.  16 flop per byte of operand traffic instead of 8: the 64 x 64 kernel above
// is L2/MALL-bandwidth bound on the large nodes of the recursion (24-39 TFLOP/s measured).
constexpr int HB = 128, HK = 16, HLD = HB + 16;
template <bool TB>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(int m, int n, int k, double alpha,
                                                          const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb,
                                                          double beta, double* __restrict__ C, int64_t ldc,
                                                          int lower_only, int tri) {
  int tm, tn;
  tile_of_block(lower_only, tri, m / HB, n / HB, tm, tn);
  __shared__ __attribute__((aligned(16))) double sm[2][2][HK][HLD];  // [stage][A|B][k][row/col]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 2, wn = w & 3;
  v4d acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  int klo = 0, khi = k;
  if (tri == 1) khi = min(k, (tn + 1) * HB);
  else if (tri == 2) klo = min(k, tn * HB);
  else if (tri == 3) khi = min(k, (tm + 1) * HB);
  else if (tri == 4) klo = min(k, max(tm, tn) * HB);
  else if (tri == 5) klo = min(k, tm * HB);

  const double* Ab = A + (int64_t)tm * HB * lda;
  const double* Bb = TB ? B + (int64_t)tn * HB * ldb : B + (int64_t)tn * HB;
  const int lr = tid >> 2, lk = (tid & 3) * 4;   // [row][k..k+3] loader (A, and B when TB)
  const int br = tid >> 5, bc = (tid & 31) * 4;  // [k][n..n+3] loader (B when !TB)
  v2d a0, a1, b0, b1;
  auto fetch = [&](int k0) {
    const double* sa = Ab + (int64_t)lr * lda + k0 + lk;
    a0 = *(const v2d*)sa;
    a1 = *(const v2d*)(sa + 2);
    const double* sb = TB ? Bb + (int64_t)lr * ldb + k0 + lk : Bb + (int64_t)(k0 + br) * ldb + bc;
    b0 = *(const v2d*)sb;
    b1 = *(const v2d*)(sb + 2);
  };
  auto stage = [&](int st) {
    double(*As)[HLD] = sm[st][0];
    double(*Bs)[HLD] = sm[st][1];
    const int sc = (lr + 2 * lk) & (HB - 1);  // column rotation by 8 (k >> 2): see gemm_kernel8
    As[lk + 0][sc] = a0.x; As[lk + 1][sc] = a0.y; As[lk + 2][sc] = a1.x; As[lk + 3][sc] = a1.y;
    if (TB) {
      Bs[lk + 0][sc] = b0.x; Bs[lk + 1][sc] = b0.y; Bs[lk + 2][sc] = b1.x; Bs[lk + 3][sc] = b1.y;
    } else {
      *(v2d*)&Bs[br][bc] = b0;
      *(v2d*)&Bs[br][bc + 2] = b1;
    }
  };
  if (klo < khi) {
    fetch(klo);
    stage(0);
  }
  __syncthreads();
  int st = 0;
  for (int k0 = klo; k0 < khi; k0 += HK) {
    const bool more = k0 + HK < khi;
    if (more) fetch(k0 + HK);
#pragma unroll
    for (int k4 = 0; k4 < HK / 4; ++k4) {
      const int kr = k4 * 4 + (lane >> 4);
      double av[4], bv[2];
#pragma unroll
      for (int f = 0; f < 4; ++f) av[f] = sm[st][0][kr][(wm * 64 + f * 16 + (lane & 15) + 8 * k4) & (HB - 1)];
#pragma unroll
      for (int f = 0; f < 2; ++f)
        bv[f] = sm[st][1][kr][TB ? (wn * 32 + f * 16 + (lane & 15) + 8 * k4) & (HB - 1) : wn * 32 + f * 16 + (lane & 15)];
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) acc[fm][fn] = mfma_f64(av[fm], bv[fn], acc[fm][fn]);
    }
    if (more) stage(st ^ 1);
    __syncthreads();
    st ^= 1;
  }
#pragma unroll
  for (int fm = 0; fm < 4; ++fm)
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = (int64_t)tm * HB + wm * 64 + fm * 16 + (lane >> 4) + 4 * r;
        const int64_t col = (int64_t)tn * HB + wn * 32 + fn * 16 + (lane & 15);
        double* dst = C + row * ldc + col;
        const double v = alpha * acc[fm][fn][r];
        *dst = (beta == 0.0) ? v : fma(beta, *dst, v);
      }
}

// C = beta C + alpha sum_z P[z]  (fixed order: deterministic) for the k-split launches below.
__global__ void ksplit_reduce_kernel(const double* __restrict__ P, int nz, int64_t m, int64_t n, int64_t ldp,
                                     double alpha, double beta, double* __restrict__ C, int64_t ldc, int k, int tri) {
  const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= m || j >= n) return;
  // the slices that met this element's 64 x 64 tile (gemm8_body's pruning, the same expressions)
  const int tm = (int)(i / GB), tn = (int)(j / GB);
  int klo = 0, khi = k;
  if (tri == 1) khi = min(k, (tn + 1) * GB);
  else if (tri == 2) klo = min(k, tn * GB);
  else if (tri == 3) khi = min(k, (tm + 1) * GB);
  else if (tri == 4) klo = min(k, max(tm, tn) * GB);
  else if (tri == 5) klo = min(k, tm * GB);
  const int per = ksplit_depth(k, nz);
  double acc = 0.0;
  for (int z = klo / per; z < nz && z * per < khi; ++z) acc += P[((int64_t)z * m + i) * ldp + j];
  double* dst = C + i * ldc + j;
  *dst = (beta == 0.0) ? alpha * acc : fma(beta, *dst, alpha * acc);
}

// Few output tiles but a long k (the gradient / cross-covariance products: N x P x N with P <= 128): one
// workgroup per tile walks a long k range alone.  Split k
// over `nz` workgroups per tile (partials in `scratch` [nz][m][n]) and reduce in a fixed order.
void launch_gemm_ksplit(hipStream_t s, bool tb, int m, int n, int k, double alpha, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int tri, int nz,
                        double* scratch) {
  dim3 grid((unsigned)((m / GB) * (n / GB)), (unsigned)nz);
  if (tb) hipLaunchKernelGGL(gemm_kernel8<true>, grid, dim3(512), 0, s, m, n, k, 1.0, A, lda, B, ldb, 0.0, scratch, (int64_t)n, 0, tri);
  else hipLaunchKernelGGL(gemm_kernel8<false>, grid, dim3(512), 0, s, m, n, k, 1.0, A, lda, B, ldb, 0.0, scratch, (int64_t)n, 0, tri);
  dim3 rg((unsigned)((n + 63) / 64), (unsigned)((m + 3) / 4));
  hipLaunchKernelGGL(ksplit_reduce_kernel, rg, dim3(256), 0, s, scratch, nz, (int64_t)m, (int64_t)n, (int64_t)n, alpha, beta, C, ldc, k, tri);
}

// k must be a multiple of 32 (all call sites pass multiples of 64).
// Measured (profiles/r01_update_breakdown.txt): a dependent kernel costs ~5 us on this part however
// small it is, and the 64x64-tile kernel is L2/MALL-bandwidth bound (8 flop per byte) on the large
// nodes of the recursion; splitting k over more waves did not change either and was dropped.
void launch_gemm(hipStream_t s, bool tb, int m, int n, int k, double alpha, const double* A,
                 int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                 bool lower_only, int tri, int small_tiles) {
  const int lo = lower_only ? 1 : 0;
  static const int64_t big_min = getenv("TGP_GEMM_BIG") ? atoll(getenv("TGP_GEMM_BIG")) : 512;  // tuning aid
  auto live_tiles = [&](int T) {  // 1-D grid over the live tiles (see tile_of_block)
    const int64_t ntm = m / T, ntn = n / T;
    return (unsigned)(lower_only ? ntm * (ntm + 1) / 2 : ntm * ntn);
  };
  // small_tiles: the caller knows a few tiles carry a very long k range (K^-1 = W^T W: the first tile
  // rows sum over all of N) -- one 128 x 128 workgroup walking k = 4096 alone on its CU is the makespan
  if (!small_tiles && m % HB == 0 && n % HB == 0 && (int64_t)(m / HB) * (n / HB) >= big_min) {
    dim3 gb(live_tiles(HB));
    if (tb) hipLaunchKernelGGL(gemm_big_kernel<true>, gb, dim3(512), 0, s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lo, tri);
    else hipLaunchKernelGGL(gemm_big_kernel<false>, gb, dim3(512), 0, s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lo, tri);
    return;
  }
  dim3 grid(live_tiles(GB));
  // (not for lower_only: the 32 x 32 form would leave the upper-right quarter of the diagonal 64 x 64 blocks
  // unwritten, and the callers of this entry point -- the NLML reduction over K^-1 -- read whole 64 x 64 tiles;
  // the factor recursion, whose consumers read strictly the lower triangle, goes through launch_node_pair)
  if (!lower_only && (int)grid.x <= small_tile_limit()) {
    dim3 gs(live_tiles(SB));
    if (tb) hipLaunchKernelGGL(gemm_small_kernel<true>, gs, dim3(512), 0, s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lo, tri);
    else hipLaunchKernelGGL(gemm_small_kernel<false>, gs, dim3(512), 0, s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lo, tri);
    return;
  }
  if (tb) hipLaunchKernelGGL(gemm_kernel8<true>, grid, dim3(512), 0, s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lo, tri);
  else hipLaunchKernelGGL(gemm_kernel8<false>, grid, dim3(512), 0, s, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, lo, tri);
}

// ---------------------------------------------------------------------------------------------
// Wt[k][i] = W[i][k] for k <= i < N, zero elsewhere (also wipes the identity the padding got).
__global__ void transpose_mask_kernel(const double* __restrict__ W, double* __restrict__ Wt, int64_t N,
                                      int64_t Npad) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.y * 32, k0 = (int64_t)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = i0 + r, k = k0 + tx;
    tile[r][tx] = (i < N && k <= i) ? W[i * Npad + k] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) Wt[(k0 + r) * Npad + i0 + tx] = tile[tx][r];
}

// dst[c][r] = src[r][c], rows x cols both multiples of 32.
__global__ void transpose_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                                 int64_t ldd) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(r0 + r) * lds + c0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) dst[(c0 + r) * ldd + r0 + tx] = tile[tx][r];
}

void launch_transpose(hipStream_t s, const double* src, int64_t rows, int64_t cols, int64_t lds, double* dst,
                      int64_t ldd) {
  dim3 grid((unsigned)(cols / 32), (unsigned)(rows / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, src, lds, dst, ldd);
}

void launch_transpose_mask(hipStream_t s, const double* W, double* Wt, int64_t N, int64_t Npad) {
  dim3 grid((unsigned)(Npad / 32), (unsigned)(Npad / 32));
  hipLaunchKernelGGL(transpose_mask_kernel, grid, dim3(256), 0, s, W, Wt, N, Npad);
}

__global__ void zero_kernel(double* p, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = 0.0;
}
void launch_zero(hipStream_t s, double* p, int64_t n) {
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

__global__ void center_kernel(const double* __restrict__ Y, double c, double* __restrict__ err,
                              int64_t N, int64_t Npad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < Npad) err[t] = (t < N) ? Y[t] - c : 0.0;
}
void launch_center(hipStream_t s, const double* Y, double c, double* err, int64_t N, int64_t Npad) {
  hipLaunchKernelGGL(center_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, s, Y, c, err, N,
                     Npad);
}

__global__ void row_norms_kernel(const double* __restrict__ Xs, double* __restrict__ xn, int64_t Npad, int dp) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Npad) return;
  double s = 0.0;
  for (int c = 0; c < dp; ++c) s = fma(Xs[t * dp + c], Xs[t * dp + c], s);
  xn[t] = s;
}
void launch_row_norms(hipStream_t s, const double* Xs, double* xn, int64_t Npad, int dp) {
  hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, s, Xs, xn, Npad, dp);
}

// one wave per row: y[i] = sum_{k in tri range} M[i][k] x[k].  The row is walked from a 16-byte aligned start
// (entries outside the triangle are explicit zeros in both users, W and Wt) with two entries per lane and load and
// four independent partial sums, so that eight loads per lane are in flight instead of one dependent chain.
__global__ __launch_bounds__(256) void trmv_kernel(const double* __restrict__ Mx, int64_t ld, int64_t n,
                                                   const double* __restrict__ x, double* __restrict__ y,
                                                   int lower) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t lo = lower ? 0 : (i & ~(int64_t)1), hi = lower ? ((i + 2) & ~(int64_t)1) : n;  // even bounds (n is even)
  const double* row = Mx + i * ld;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t k = lo + 2 * lane;
  for (; k + 384 < hi; k += 512) {
    const v2d m0 = *(const v2d*)(row + k), m1 = *(const v2d*)(row + k + 128);
    const v2d m2 = *(const v2d*)(row + k + 256), m3 = *(const v2d*)(row + k + 384);
    const v2d x0 = *(const v2d*)(x + k), x1 = *(const v2d*)(x + k + 128);
    const v2d x2 = *(const v2d*)(x + k + 256), x3 = *(const v2d*)(x + k + 384);
    a0 = fma(m0.y, x0.y, fma(m0.x, x0.x, a0));
    a1 = fma(m1.y, x1.y, fma(m1.x, x1.x, a1));
    a2 = fma(m2.y, x2.y, fma(m2.x, x2.x, a2));
    a3 = fma(m3.y, x3.y, fma(m3.x, x3.x, a3));
  }
  for (; k < hi; k += 128) {
    const v2d m0 = *(const v2d*)(row + k), x0 = *(const v2d*)(x + k);
    a0 = fma(m0.y, x0.y, fma(m0.x, x0.x, a0));
  }
  double acc = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) y[i] = acc;
}
void launch_trmv(hipStream_t s, const double* Mx, int64_t ld, int64_t n, const double* x, double* y,
                 bool lower) {
  hipLaunchKernelGGL(trmv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, Mx, ld, n, x, y,
                     lower ? 1 : 0);
}

__global__ void axpby_kernel(int64_t n, double a, const double* x, double b, const double* y,
                             double* out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = a * (x ? x[t] : 0.0) + b * (y ? y[t] : 0.0);
}
void launch_axpby_vec(hipStream_t s, int64_t n, double a, const double* x, double b, const double* y,
                      double* out) {
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a, x, b, y,
                     out);
}

}  // namespace tgp
