// `update` path kernels for gfx950: input scaling, K assembly, 64x64 Cholesky+inverse leaf,
// f64-MFMA GEMM (the trailing updates of the recursive Cholesky / triangular inverse),
// masked transpose, triangular mat-vec.
//
// Replaces (reference call sites): self.model.kernel(x) and tf.linalg.cholesky(K + s) inside
// gpflow GPRPosterior._precompute reached from trieste/models/gpflow/models.py:171-186 ->
// interface.py:108-112.  
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

// K[i][j] = k(x_i, x_j) + noise * (i == j) on the N x N part; identity on the padding so that the
// padded matrix stays SPD and factorises to blockdiag(L, I).
__global__ void assemble_K_kernel(const double* __restrict__ Xs, double* __restrict__ A, int64_t N,
                                  int64_t Npad, int dp, int kind, double variance, double noise) {
  const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int64_t i = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= Npad || j >= Npad) return;
  double v;
  if (i < N && j < N) {
    double r2 = 0.0;
    for (int c = 0; c < dp; ++c) {
      const double t = Xs[i * dp + c] - Xs[j * dp + c];
      r2 = fma(t, t, r2);
    }
    v = kernel_rt(kind, r2, variance);
    if (i == j) v += noise;
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  A[i * Npad + j] = v;
}

void launch_assemble_K(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp,
                       int kind, double variance, double noise) {
  dim3 grid((unsigned)(Npad / 16), (unsigned)(Npad / 16));
  hipLaunchKernelGGL(assemble_K_kernel, grid, dim3(256), 0, s, Xs, A, N, Npad, dp, kind, variance,
                     noise);
}

// ---------------------------------------------------------------------------------------------
// Leaf: Cholesky of a 64x64 diagonal block and the inverse of its factor, one workgroup.
// info: 0 = ok, else 1 + global index of the first non-positive pivot.
__global__ __launch_bounds__(256) void leaf_kernel(const double* __restrict__ A, double* __restrict__ L,
                                                   double* __restrict__ W, int64_t ld, int64_t off,
                                                   int* __restrict__ info) {
  __shared__ double S[LEAF][LEAF + 1];
  __shared__ double T[LEAF][LEAF + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < LEAF * LEAF; e += 256) {
    const int i = e >> 6, j = e & 63;
    S[i][j] = (j <= i) ? A[(off + i) * ld + off + j] : 0.0;
    T[i][j] = 0.0;
  }
  for (int j = 0; j < LEAF; ++j) {
    __syncthreads();  // loads / trailing update of step j-1 complete
    double dj = S[j][j];
    if (!(dj > 0.0)) {  // also catches NaN
      if (tid == 0) atomicCAS(info, 0, (int)(off + j) + 1);
      dj = 1.0;
    }
    const double sd = sqrt(dj);
    __syncthreads();
    if (tid < LEAF) {
      if (tid == j) S[j][j] = sd;
      else if (tid > j) S[tid][j] = S[tid][j] / sd;
    }
    __syncthreads();
    // trailing update of the lower triangle: S[i][k] -= S[i][j] * S[k][j], j < k <= i
    const int i = tid >> 2;
    if (i > j) {
      const double lij = S[i][j];
      for (int k = j + 1 + (tid & 3); k <= i; k += 4) S[i][k] = fma(-lij, S[k][j], S[i][k]);
    }
  }
  __syncthreads();
  // T = S^-1 by forward substitution, one column per lane of wave 0.
  if (tid < LEAF) {
    const int c = tid;
    for (int i = 0; i < LEAF; ++i) {
      double acc = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) {
        const double t = (k >= c) ? T[k][c] : 0.0;
        acc = fma(-S[i][k], t, acc);
      }
      if (i >= c) T[i][c] = acc / S[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < LEAF * LEAF; e += 256) {
    const int i = e >> 6, j = e & 63;
    L[(off + i) * ld + off + j] = (j <= i) ? S[i][j] : 0.0;
    W[(off + i) * ld + off + j] = (j <= i) ? T[i][j] : 0.0;
  }
}

void launch_leaf(hipStream_t s, const double* A, double* L, double* W, int64_t ld, int64_t off,
                 int* info) {
  hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), 0, s, A, L, W, ld, off, info);
}

// ---------------------------------------------------------------------------------------------
// f64 MFMA GEMM, 64x64 tile per workgroup, 4 waves (2x2), each wave 32x32 = 2x2 fragments of
// v_mfma_f64_16x16x4_f64.  m, n multiples of 64; k multiple of 16.  Operand tiles are staged in
// LDS k-major ([k][m] / [k][n]) with a +16-double row pad so that the 4 k-rows a fragment read
// touches fall on disjoint bank halves.
constexpr int GB = 64, GK = 16, GLD = GB + 16;

template <bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(int m, int n, int k, double alpha,
                                                   const double* __restrict__ A, int64_t lda,
                                                   const double* __restrict__ B, int64_t ldb,
                                                   double beta, double* __restrict__ C, int64_t ldc,
                                                   int lower_only) {
  const int tn = blockIdx.x, tm = blockIdx.y;
  if (lower_only && tn > tm) return;
  __shared__ double As[GK][GLD];
  __shared__ double Bs[GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  const double* Ab = A + (int64_t)tm * GB * lda;
  const double* Bb = TB ? B + (int64_t)tn * GB * ldb : B + (int64_t)tn * GB;
  const int lr = tid >> 2, lk = (tid & 3) * 4;       // [row][k..k+3] loader (A, and B when TB)
  const int br = tid >> 4, bc = (tid & 15) * 4;      // [k][n..n+3] loader (B when !TB)

  for (int k0 = 0; k0 < k; k0 += GK) {
    {
      const double* src = Ab + (int64_t)lr * lda + k0 + lk;
      const v2d x0 = *(const v2d*)src, x1 = *(const v2d*)(src + 2);
      As[lk + 0][lr] = x0.x; As[lk + 1][lr] = x0.y; As[lk + 2][lr] = x1.x; As[lk + 3][lr] = x1.y;
    }
    if (TB) {
      const double* src = Bb + (int64_t)lr * ldb + k0 + lk;
      const v2d x0 = *(const v2d*)src, x1 = *(const v2d*)(src + 2);
      Bs[lk + 0][lr] = x0.x; Bs[lk + 1][lr] = x0.y; Bs[lk + 2][lr] = x1.x; Bs[lk + 3][lr] = x1.y;
    } else {
      const double* src = Bb + (int64_t)(k0 + br) * ldb + bc;
      const v2d x0 = *(const v2d*)src, x1 = *(const v2d*)(src + 2);
      *(v2d*)&Bs[br][bc] = x0;
      *(v2d*)&Bs[br][bc + 2] = x1;
    }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < GK / 4; ++k4) {
      const int kr = k4 * 4 + (lane >> 4);
      double a[2], b[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        a[f] = As[kr][wm * 32 + f * 16 + (lane & 15)];
        b[f] = Bs[kr][wn * 32 + f * 16 + (lane & 15)];
      }
#pragma unroll
      for (int fa = 0; fa < 2; ++fa)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) acc[fa][fb] = mfma_f64(a[fa], b[fb], acc[fa][fb]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int fa = 0; fa < 2; ++fa)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = (int64_t)tm * GB + wm * 32 + fa * 16 + (lane >> 4) + 4 * r;
        const int64_t col = (int64_t)tn * GB + wn * 32 + fb * 16 + (lane & 15);
        double* dst = C + row * ldc + col;
        const double v = alpha * acc[fa][fb][r];
        *dst = (beta == 0.0) ? v : fma(beta, *dst, v);
      }
}

void launch_gemm(hipStream_t s, bool tb, int m, int n, int k, double alpha, const double* A,
                 int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                 bool lower_only) {
  dim3 grid((unsigned)(n / GB), (unsigned)(m / GB));
  if (tb)
    hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, m, n, k, alpha, A, lda, B, ldb, beta,
                       C, ldc, lower_only ? 1 : 0);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, m, n, k, alpha, A, lda, B, ldb, beta,
                       C, ldc, lower_only ? 1 : 0);
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

// one wave per row: y[i] = sum_{k in tri range} M[i][k] x[k]
__global__ __launch_bounds__(256) void trmv_kernel(const double* __restrict__ Mx, int64_t ld, int64_t n,
                                                   const double* __restrict__ x, double* __restrict__ y,
                                                   int lower) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const int64_t lo = lower ? 0 : i, hi = lower ? i + 1 : n;
  double acc = 0.0;
  for (int64_t k = lo + lane; k < hi; k += 64) acc = fma(Mx[i * ld + k], x[k], acc);
  acc = wave_sum(acc);
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
