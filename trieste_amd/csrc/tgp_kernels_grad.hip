// Acquisition value AND gradient w.r.t. the query point, for the L-BFGS-B refinement of the sweep
// winners (reference trieste/acquisition/optimizer.py:344-745: generate_continuous_optimizer /
// _perform_parallel_continuous_optimization, where the gradient comes from
// tfp.math.value_and_gradient at :628-629).  Analytic form of the same derivative:
//     d mean / dx = sum_k alpha_k dk_k/dx,     d var / dx = -2 sum_k z_k dk_k/dx,   z = K^-1 k*
// with z = W^T (W k*) from the cached inverse factor (two f64-MFMA GEMMs over the P query points),
// then the chain rule through EI / PI / -LCB.  P is small (tens to hundreds of L-BFGS-B iterates).
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

namespace tgp {

// B[k][p] = k(X_k, x_p), k-major [Npad][Ppad], zero padded.
__global__ void kstar_t_kernel(ModelDev m, const double* __restrict__ Xq, int64_t P, int64_t Ppad,
                               double* __restrict__ B) {
  const int64_t p = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t k = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (p >= Ppad || k >= m.Npad) return;
  double v = 0.0;
  if (p < P && k < m.N) {
    double r2 = 0.0;
    for (int c = 0; c < m.d; ++c) {
      const double t = Xq[p * m.d + c] / m.ls[c] - m.Xs[k * m.dp + c];
      r2 = fma(t, t, r2);
    }
    v = kernel_rt(m.kind, r2, m.variance);
  }
  B[k * Ppad + p] = v;
}

// out[i][j] = k(x1_i, x2_j) - S[i][j]  (covariance_between_points, reference models.py:188-254)
__global__ void cov_tail_kernel(ModelDev m, const double* __restrict__ X1, int64_t P1,
                                const double* __restrict__ X2, int64_t P2, const double* __restrict__ S,
                                int64_t lds, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= P1 || j >= P2) return;
  double r2 = 0.0;
  for (int c = 0; c < m.d; ++c) {
    const double t = (X1[i * m.d + c] - X2[j * m.d + c]) / m.ls[c];
    r2 = fma(t, t, r2);
  }
  out[i * P2 + j] = kernel_rt(m.kind, r2, m.variance) - S[i * lds + j];
}

void launch_cov_tail(hipStream_t s, const ModelDev& m, const double* X1, int64_t P1, const double* X2,
                     int64_t P2, const double* S, int64_t lds, double* out) {
  dim3 grid((unsigned)((P2 + 63) / 64), (unsigned)((P1 + 3) / 4));
  hipLaunchKernelGGL(cov_tail_kernel, grid, dim3(256), 0, s, m, X1, P1, X2, P2, S, lds, out);
}

// A[i][j] (lower triangle incl. diagonal, mirrored to the upper) = k(x_i, x_j) - S[i][j] + jitter [i == j]
// for i, j < n; the padding carries the identity so the padded matrix stays positive definite.
__global__ void cov_sym_tail_kernel(ModelDev m, const double* __restrict__ X, int64_t n, int64_t Pp,
                                    const double* __restrict__ S, double jitter, double* __restrict__ A) {
  const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= Pp || j >= Pp) return;
  double v;
  if (i < n && j < n) {
    double r2 = 0.0;
    for (int c = 0; c < m.d; ++c) {
      const double t = (X[i * m.d + c] - X[j * m.d + c]) / m.ls[c];
      r2 = fma(t, t, r2);
    }
    const int64_t a = i > j ? i : j, b = i > j ? j : i;  // S holds the lower triangle
    v = kernel_rt(m.kind, r2, m.variance) - S[a * Pp + b] + (i == j ? jitter : 0.0);
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  A[i * Pp + j] = v;
}

void launch_cov_sym_tail(hipStream_t s, const ModelDev& m, const double* X, int64_t n, int64_t Pp, const double* S,
                         double jitter, double* A) {
  dim3 grid((unsigned)(Pp / 64), (unsigned)(Pp / 4));
  hipLaunchKernelGGL(cov_sym_tail_kernel, grid, dim3(256), 0, s, m, X, n, Pp, S, jitter, A);
}

// E [rp, cp] = src [r, c] zero padded
__global__ void pad_copy_kernel(const double* __restrict__ src, int64_t r, int64_t c, double* __restrict__ dst,
                                int64_t rp, int64_t cp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rp * cp) return;
  const int64_t i = e / cp, j = e % cp;
  dst[e] = (i < r && j < c) ? src[i * c + j] : 0.0;
}

void launch_pad_copy(hipStream_t s, const double* src, int64_t r, int64_t c, double* dst, int64_t rp, int64_t cp) {
  hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((rp * cp + 255) / 256)), dim3(256), 0, s, src, r, c, dst, rp, cp);
}

// out [S, n]: out[s][j] = mean[j] + R[j][s]
__global__ void sample_tail_kernel(const double* __restrict__ mean, const double* __restrict__ R, int64_t n, int S,
                                   int64_t Sp, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)S * n) return;
  const int64_t sidx = e / n, j = e % n;
  out[e] = mean[j] + R[j * Sp + sidx];
}

void launch_sample_tail(hipStream_t s, const double* mean, const double* R, int64_t n, int S, int64_t Sp, double* out) {
  hipLaunchKernelGGL(sample_tail_kernel, dim3((unsigned)(((int64_t)S * n + 255) / 256)), dim3(256), 0, s, mean, R, n,
                     S, Sp, out);
}

// A[i][j] <- (negate ? -1 : 1) * scale * A[i][j] + shift [i == j]  for i, j < n (lower triangle is the
// input, the result is mirrored);  padding rows/cols: identity.  Finishes D = Phi^T Phi + noise I,
// G = Phi Phi^T + noise I, noise * D^-1 and I - A^T A of the RFF weight posterior.
__global__ void sym_finish_kernel(double* __restrict__ A, int64_t n, int64_t np, double scale, double shift,
                                  int negate) {
  const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= np || j > i) return;
  double v;
  if (i < n) {
    v = scale * A[i * np + j];
    if (negate) v = -v;
    if (i == j) v += shift;
  } else {
    v = (i == j) ? 1.0 : 0.0;
  }
  A[i * np + j] = v;
  A[j * np + i] = v;
}

void launch_sym_finish(hipStream_t s, double* A, int64_t n, int64_t np, double scale, double shift, int negate) {
  dim3 grid((unsigned)(np / 64), (unsigned)(np / 4));
  hipLaunchKernelGGL(sym_finish_kernel, grid, dim3(256), 0, s, A, n, np, scale, shift, negate);
}

// theta[f][b] = mean[f] + R[f][b];  ws[f][b] = feature scale * theta  (what the trajectory kernels consume)
__global__ void theta_tail_kernel(const double* __restrict__ mean, int64_t ldm, const double* __restrict__ R,
                                  int64_t ldr, int F, int B, double scale, double* __restrict__ theta,
                                  double* __restrict__ ws) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)F * B) return;
  const int64_t f = e / B, b = e % B;
  const double th = mean[f * ldm] + R[f * ldr + b];
  theta[e] = th;
  ws[e] = scale * th;
}

void launch_theta_tail(hipStream_t s, const double* mean, int64_t ldm, const double* R, int64_t ldr, int F, int B,
                       double scale, double* theta, double* ws) {
  hipLaunchKernelGGL(theta_tail_kernel, dim3((unsigned)(((int64_t)F * B + 255) / 256)), dim3(256), 0, s, mean, ldm, R,
                     ldr, F, B, scale, theta, ws);
}

constexpr int GT_THREADS = 256, GT_KS = 16, GT_J = 2 + 2 * MAX_D;
size_t grad_tail_scratch_doubles(int64_t Ppad) { return (size_t)GT_KS * (size_t)Ppad * GT_J; }

// Value and gradient of the acquisition function at P points from B = K*^T [N][Ppad], C1 = W K*, Z = K^-1 k* (same layout):
//   mean = sum_k B alpha,  var = s_f^2 - sum_k C1^2,  d mean / dx = sum_k alpha dk/dx,  d var / dx = -2 sum_k Z dk/dx.
// Two passes since round 6 (one workgroup per point before: its lanes walked k, i.e. a column of the three row-major arrays --
// 3 N cache lines of 8 useful bytes per point, 113 us at N = 4096, P = 80 on 80 of the 256 compute units):
//   (1) grad_partial_kernel, grid (Ppad / 16, GT_KS): a workgroup takes 16 points x 1/16 of the rows; lane = (point, row
//       phase), so a wave's loads are four rows of 128 contiguous bytes; partial sums -> part[ks][p][j], every sum in a fixed order;
//   (2) grad_finish_kernel: one thread per point adds the GT_KS partials in order and evaluates the acquisition's tail.
// (DP = the padded input dimension: the per-lane arrays are indexed by unrolled loops only -- with the run-time d of round 5's kernel
// they lived in scratch)
template <int DP>
__global__ __launch_bounds__(GT_THREADS) void grad_partial_kernel(ModelDev m, const double* __restrict__ Xq, int64_t P,
                                                                  int64_t Ppad, const double* __restrict__ B,
                                                                  const double* __restrict__ C1,
                                                                  const double* __restrict__ Z,
                                                                  double* __restrict__ part) {
  __shared__ double red[4][16][GT_J + 1];
  const int d = m.d, tid = threadIdx.x, pl = tid & 15, kl = tid >> 4, w = tid >> 6, lane = tid & 63;
  const int64_t p = 16 * (int64_t)blockIdx.x + pl, pc = p < P ? p : P - 1;   // (padded columns repeat the last point: never stored)
  const int ks = blockIdx.y;
  const int64_t chunk = (m.Npad + GT_KS - 1) / GT_KS;
  const int64_t k0 = ks * chunk, k1 = (k0 + chunk < m.N) ? k0 + chunk : m.N;
  // (x / ls by DIVISION, as Xs was formed: a query point that IS a training point must give t = 0 exactly -- the kernels with a kink
  // at r = 0 divide by r)
  double xs[DP], ls[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) {
    ls[c] = m.ls[c];                               // (padded with 1.0)
    xs[c] = c < d ? Xq[pc * d + c] / ls[c] : 0.0;  // (Xs is zero padded)
  }
  double mean = 0.0, ssq = 0.0;
  double gm[DP], gv[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) gm[c] = gv[c] = 0.0;
  for (int64_t k = k0 + kl; k < k1; k += 16) {
    const double kv = B[k * Ppad + p], ck = C1[k * Ppad + p], zk = Z[k * Ppad + p];
    const double al = m.alpha[k];
    mean = fma(kv, al, mean);
    ssq = fma(ck, ck, ssq);
    double t[DP], r2 = 0.0;
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      t[c] = xs[c] - m.Xs[k * DP + c];
      r2 = fma(t[c], t[c], r2);
    }
    const double f1 = 2.0 * kernel_dr2(m.kind, r2, m.variance);
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      const double dk = f1 * t[c] / ls[c];
      gm[c] = fma(al, dk, gm[c]);
      gv[c] = fma(zk, dk, gv[c]);
    }
  }
  // the four row phases of a wave (lanes pl, pl + 16, pl + 32, pl + 48), then the four waves through LDS: a fixed order
  auto fold = [&](double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
  };
  mean = fold(mean);
  ssq = fold(ssq);
#pragma unroll
  for (int c = 0; c < DP; ++c) {
    gm[c] = fold(gm[c]);
    gv[c] = fold(gv[c]);
  }
  if (lane < 16) {
    red[w][pl][0] = mean;
    red[w][pl][1] = ssq;
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      red[w][pl][2 + c] = gm[c];
      red[w][pl][2 + MAX_D + c] = gv[c];
    }
  }
  __syncthreads();
  if (tid < 16) {
    double* const out = part + ((size_t)ks * Ppad + p) * GT_J;
    auto tot = [&](int j) { return (red[0][pl][j] + red[1][pl][j]) + (red[2][pl][j] + red[3][pl][j]); };
    out[0] = tot(0);
    out[1] = tot(1);
    for (int c = 0; c < d; ++c) {
      out[2 + c] = tot(2 + c);
      out[2 + MAX_D + c] = tot(2 + MAX_D + c);
    }
  }
}

// one workgroup per point: thread j adds the GT_KS partials of sum j in order (coalesced over j), thread 0 evaluates the tail
__global__ __launch_bounds__(128) void grad_finish_kernel(ModelDev m, int64_t P, int64_t Ppad,
                                                          const double* __restrict__ part, int acq, double param,
                                                          const double* __restrict__ samples, int S, double rep_w,
                                                          double accum, double* __restrict__ val,
                                                          double* __restrict__ grad) {
  __shared__ double sums[GT_J];
  const int64_t p = blockIdx.x;
  const int d = m.d, j = threadIdx.x;
  if (j < GT_J) {
    double s = 0.0;
    for (int ks = 0; ks < GT_KS; ++ks) s += part[((size_t)ks * Ppad + p) * GT_J + j];
    sums[j] = s;
  }
  __syncthreads();
  auto tot = [&](int jj) { return sums[jj]; };
  if (j == 0) {
    const double mu = tot(0) + m.mean_const;
    const double var_raw = m.variance - tot(1);
    const bool clipped = !(var_raw > VAR_FLOOR);
    const double var = clipped ? VAR_FLOOR : var_raw;
    const double sd = sqrt(var);
    double v, dv_dmu, dv_dvar;
    if (acq == ACQ_EI || acq == ACQ_AEI) {
      const double z = (param - mu) / sd;
      const double cdf = normal_cdf(z), pdf = normal_pdf(z);
      v = (param - mu) * cdf + sd * pdf;
      dv_dmu = -cdf;
      dv_dvar = pdf / (2.0 * sd);
      if (acq == ACQ_AEI) {  // v = EI * aug(var), aug = 1 - sqrt(noise) / sqrt(noise + var)
        const double sn = sqrt(m.noise), st = sqrt(m.noise + var);
        const double aug = 1.0 - sn / st;
        dv_dvar = dv_dvar * aug + v * (0.5 * sn / (st * (m.noise + var)));
        dv_dmu *= aug;
        v *= aug;
      }
    } else if (acq == ACQ_PI) {
      const double z = (param - mu) / sd;
      const double pdf = normal_pdf(z);
      v = normal_cdf(z);
      dv_dmu = -pdf / sd;
      dv_dvar = -pdf * z / (2.0 * var);
    } else if (acq >= ACQ_MES) {  // entropy tails; GIBBON's repulsion: this model's half, -w/2 log(var + noise)
      entropy_tail(acq, mu, var, m.noise, samples, S, v, dv_dmu, dv_dvar);
      if (rep_w != 0.0) {
        v -= 0.5 * rep_w * log(var + m.noise);
        dv_dvar -= 0.5 * rep_w / (var + m.noise);
      }
    } else {
      v = -(mu - param * sd);
      dv_dmu = -1.0;
      dv_dvar = param / (2.0 * sd);
    }
    // accum != 0: add accum * (value, gradient) to what is there (the conditioned twin's half of the repulsion)
    val[p] = accum != 0.0 ? val[p] + accum * v : v;
    for (int c = 0; c < d; ++c) {
      const double dmu = tot(2 + c);
      const double dvar = clipped ? 0.0 : -2.0 * tot(2 + MAX_D + c);  // clip_by_value has zero gradient
      const double g = dv_dmu * dmu + dv_dvar * dvar;
      grad[p * d + c] = accum != 0.0 ? grad[p * d + c] + accum * g : g;
    }
  }
}

// ---- predict at a handful of points (round 6) --------------------------------------------------------------------------------
// mean = sum_k K*[k][p] alpha_k + c,  var = max(s_f^2 - sum_k (W K*)[k][p]^2, floor)  from B = K*^T and C1 = W K* ([N][Ppad], the
// arrays of the value-and-gradient call): the same two passes as the gradient tail, two sums per point.  The sweep kernels are
// built for 10^4 .. 10^7 candidates; a workgroup of theirs walks all of W for its 64 candidates -- 1.7 ms at N = 4096 for ONE point.
__global__ __launch_bounds__(GT_THREADS) void predict_small_partial_kernel(const double* __restrict__ B,
                                                                           const double* __restrict__ C1,
                                                                           const double* __restrict__ alpha, int64_t N,
                                                                           int64_t Npad, int64_t Ppad, int with_var,
                                                                           double* __restrict__ part) {
  __shared__ double red[4][16][2];
  const int tid = threadIdx.x, pl = tid & 15, kl = tid >> 4, w = tid >> 6, lane = tid & 63;
  const int64_t p = 16 * (int64_t)blockIdx.x + pl;
  const int ks = blockIdx.y;
  const int64_t chunk = (Npad + GT_KS - 1) / GT_KS;
  const int64_t k0 = ks * chunk, k1 = (k0 + chunk < N) ? k0 + chunk : N;
  double mean = 0.0, ssq = 0.0;
  for (int64_t k = k0 + kl; k < k1; k += 16) {
    mean = fma(B[k * Ppad + p], alpha[k], mean);
    if (with_var) {
      const double ck = C1[k * Ppad + p];
      ssq = fma(ck, ck, ssq);
    }
  }
  mean += __shfl_xor(mean, 16);
  mean += __shfl_xor(mean, 32);
  ssq += __shfl_xor(ssq, 16);
  ssq += __shfl_xor(ssq, 32);
  if (lane < 16) {
    red[w][pl][0] = mean;
    red[w][pl][1] = ssq;
  }
  __syncthreads();
  if (tid < 16) {
    double* const out = part + ((size_t)ks * Ppad + p) * 2;
    out[0] = (red[0][pl][0] + red[1][pl][0]) + (red[2][pl][0] + red[3][pl][0]);
    out[1] = (red[0][pl][1] + red[1][pl][1]) + (red[2][pl][1] + red[3][pl][1]);
  }
}

__global__ __launch_bounds__(64) void predict_small_finish_kernel(const double* __restrict__ part, int64_t P, int64_t Ppad,
                                                                  double variance, double mean_const,
                                                                  double* __restrict__ mean_out,
                                                                  double* __restrict__ var_out) {
  const int64_t p = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (p >= P) return;
  double mu = 0.0, s = 0.0;
  for (int ks = 0; ks < GT_KS; ++ks) {
    mu += part[((size_t)ks * Ppad + p) * 2];
    s += part[((size_t)ks * Ppad + p) * 2 + 1];
  }
  if (mean_out) mean_out[p] = mu + mean_const;
  if (var_out) var_out[p] = fmax(variance - s, VAR_FLOOR);   // reference interface.py:123
}

size_t predict_small_scratch_doubles(int64_t Ppad) { return (size_t)GT_KS * (size_t)Ppad * 2; }
void launch_predict_small_tail(hipStream_t s, const ModelDev& m, int64_t P, int64_t Ppad, const double* B, const double* C1,
                               double* part, double* mean_out, double* var_out) {
  hipLaunchKernelGGL(predict_small_partial_kernel, dim3((unsigned)(Ppad / 16), (unsigned)GT_KS), dim3(GT_THREADS), 0, s, B, C1,
                     m.alpha, m.N, m.Npad, Ppad, var_out ? 1 : 0, part);
  hipLaunchKernelGGL(predict_small_finish_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, s, part, P, Ppad, m.variance,
                     m.mean_const, mean_out, var_out);
}

void launch_kstar_t(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, double* B) {
  dim3 grid((unsigned)(Ppad / 64), (unsigned)(m.Npad / 4));
  hipLaunchKernelGGL(kstar_t_kernel, grid, dim3(256), 0, s, m, Xq, P, Ppad, B);
}

// ---- vector-Jacobian product of predict_joint (round 6): the gradient of qEI for the L-BFGS-B refinement ----------------------
// phi = sum_i gm_i mean_i + sum_ij gc_ij cov_ij over a group of q points (mean_i = k*_i^T alpha + c, cov_ij = k(x_i, x_j) - c_i^T c_j,
// c_i = W k*_i; reference interface.py:126-133).  With Gs = gc + gc^T:
//     d phi / d x_i = (d k*_i / d x_i)^T [gm_i alpha - W^T sum_j Gs_ij c_j] + sum_{j != i} Gs_ij d k(x_i, x_j) / d x_i
// -- what TF autodiff gives the reference's optimizer (optimizer.py:628-629) through predict_joint, in analytic form.
// (1) joint_mix_kernel: D[k][p] = sum_{j in group(p)} Gs[p][j] C1[k][j]  (k-major [Npad][Ppad] like C1; padded columns zero);
// (2) Z' = W^T D by the tall GEMM; (3) grad_partial_kernel with Z' in the place of Z: part[2 + c] = sum_k alpha_k dk_k/dx_c,
// part[2 + MAX_D + c] = sum_k Z'_k dk_k/dx_c; (4) joint_vjp_finish_kernel adds the partials in order and the group's cross terms.
__global__ void joint_mix_kernel(const double* __restrict__ C1, const double* __restrict__ gcov, int64_t P, int64_t Ppad,
                                 int64_t Npad, int q, double* __restrict__ D) {
  const int64_t p = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t k = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (p >= Ppad || k >= Npad) return;
  double v = 0.0;
  if (p < P) {
    const int64_t g = p / q, i = p - g * q;
    const double* const G = gcov + (size_t)g * q * q;
    const double* const row = C1 + k * Ppad + g * q;
    for (int j = 0; j < q; ++j) v = fma(G[i * q + j] + G[j * q + i], row[j], v);
  }
  D[k * Ppad + p] = v;
}

__global__ __launch_bounds__(64) void joint_vjp_finish_kernel(ModelDev m, const double* __restrict__ Xq, int64_t P,
                                                              int64_t Ppad, int q, const double* __restrict__ part,
                                                              const double* __restrict__ gmean,
                                                              const double* __restrict__ gcov, double* __restrict__ grad) {
  const int64_t p = blockIdx.x;
  const int d = m.d, c = threadIdx.x;
  if (c >= d) return;
  double sa = 0.0, sz = 0.0;
  for (int ks = 0; ks < GT_KS; ++ks) {
    sa += part[((size_t)ks * Ppad + p) * GT_J + 2 + c];
    sz += part[((size_t)ks * Ppad + p) * GT_J + 2 + MAX_D + c];
  }
  double g = gmean[p] * sa - sz;
  const int64_t grp = p / q, i = p - grp * q;
  const double* const G = gcov + (size_t)grp * q * q;
  for (int j = 0; j < q; ++j) {
    if (j == i) continue;   // k(x, x) = s_f^2: no dependence on x
    const int64_t pj = grp * q + j;
    double r2 = 0.0, tc = 0.0;
    for (int cc = 0; cc < d; ++cc) {
      const double t = (Xq[p * d + cc] - Xq[pj * d + cc]) / m.ls[cc];
      r2 = fma(t, t, r2);
      if (cc == c) tc = t;
    }
    const double f1 = 2.0 * kernel_dr2(m.kind, r2, m.variance);
    g = fma(G[i * q + j] + G[j * q + i], f1 * tc / m.ls[c], g);
  }
  grad[p * d + c] = g;
}

// cov[g][i][j] = k(x_i, x_j) - S[g q + i][g q + j], the diagonal clipped at the floor (reference interface.py:126-133); S = C1^T C1
// over all the call's points, of which the q x q diagonal blocks are the groups' (the off-diagonal blocks are not read)
__global__ void joint_pick_kernel(ModelDev m, const double* __restrict__ Xq, int64_t P, int64_t Ppad, int q,
                                  const double* __restrict__ S, double* __restrict__ cov) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= P * q) return;
  const int64_t p = e / q, j = e - p * q, g = p / q, pj = g * q + j;
  double v;
  if (pj == p) {
    v = fmax(m.variance - S[p * Ppad + p], VAR_FLOOR);
  } else {
    double r2 = 0.0;
    for (int c = 0; c < m.d; ++c) {
      const double t = (Xq[p * m.d + c] - Xq[pj * m.d + c]) / m.ls[c];
      r2 = fma(t, t, r2);
    }
    // (the product's two triangles agree to rounding only: read the lower one for both, as the joint kernel's halves are identical)
    const int64_t a = p > pj ? p : pj, b = p > pj ? pj : p;
    v = kernel_rt(m.kind, r2, m.variance) - S[a * Ppad + b];
  }
  cov[e] = v;
}

void launch_joint_pick(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, int q, const double* S,
                       double* cov) {
  hipLaunchKernelGGL(joint_pick_kernel, dim3((unsigned)((P * q + 255) / 256)), dim3(256), 0, s, m, Xq, P, Ppad, q, S, cov);
}

void launch_joint_mix(hipStream_t s, const double* C1, const double* gcov, int64_t P, int64_t Ppad, int64_t Npad, int q,
                      double* D) {
  dim3 grid((unsigned)(Ppad / 64), (unsigned)(Npad / 4));
  hipLaunchKernelGGL(joint_mix_kernel, grid, dim3(256), 0, s, C1, gcov, P, Ppad, Npad, q, D);
}

static void launch_grad_partial(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, const double* B,
                                const double* C1, const double* Z, double* part);

void launch_joint_vjp_tail(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, int q, const double* B,
                           const double* C1, const double* Z, double* part, const double* gmean, const double* gcov,
                           double* grad) {
  launch_grad_partial(s, m, Xq, P, Ppad, B, C1, Z, part);
  hipLaunchKernelGGL(joint_vjp_finish_kernel, dim3((unsigned)P), dim3(64), 0, s, m, Xq, P, Ppad, q, part, gmean, gcov, grad);
}

static void launch_grad_partial(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, const double* B,
                                const double* C1, const double* Z, double* part) {
  const dim3 grid((unsigned)(Ppad / 16), (unsigned)GT_KS);
  if (m.dp == 2) hipLaunchKernelGGL(grad_partial_kernel<2>, grid, dim3(GT_THREADS), 0, s, m, Xq, P, Ppad, B, C1, Z, part);
  else if (m.dp == 4) hipLaunchKernelGGL(grad_partial_kernel<4>, grid, dim3(GT_THREADS), 0, s, m, Xq, P, Ppad, B, C1, Z, part);
  else if (m.dp == 6) hipLaunchKernelGGL(grad_partial_kernel<6>, grid, dim3(GT_THREADS), 0, s, m, Xq, P, Ppad, B, C1, Z, part);
  else if (m.dp == 8) hipLaunchKernelGGL(grad_partial_kernel<8>, grid, dim3(GT_THREADS), 0, s, m, Xq, P, Ppad, B, C1, Z, part);
  else if (m.dp == 16) hipLaunchKernelGGL(grad_partial_kernel<16>, grid, dim3(GT_THREADS), 0, s, m, Xq, P, Ppad, B, C1, Z, part);
  else hipLaunchKernelGGL(grad_partial_kernel<32>, grid, dim3(GT_THREADS), 0, s, m, Xq, P, Ppad, B, C1, Z, part);
}

void launch_grad_tail(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad,
                      const double* B, const double* C1, const double* Z, double* part, int acq, double param, double* val,
                      double* grad, const double* samples, int S, double rep_w, double accum) {
  launch_grad_partial(s, m, Xq, P, Ppad, B, C1, Z, part);
  hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)P), dim3(128), 0, s, m, P, Ppad, part, acq, param,
                     samples, S, rep_w, accum, val, grad);
}

}  // namespace tgp

// =============================================================================================
// Negative log marginal likelihood and its gradient w.r.t. (lengthscales[d], variance, noise, mean)
// -- the loss gpflow.optimizers.Scipy minimises inside GaussianProcessRegression.optimize_encoded
// (reference trieste/models/gpflow/models.py:256-292; gpflow GPR.training_loss = -log p(y) - log
// prior).  The engine provides the likelihood part:
//     nlml = 1/2 err^T alpha + sum_i log L_ii + N/2 log(2 pi)
//     d nlml / d theta = 1/2 sum_ij (Kinv_ij - alpha_i alpha_j) dK_ij / d theta,   Kinv = W^T W
// (the reference differentiates the Cholesky-based expression by TF autodiff; same derivative).
namespace tgp {

constexpr int NG_MAXP = MAX_D + 2;  // d lengthscales + variance + noise

// One workgroup per 64x64 tile of (i, j) pairs: lane -> j (coalesced Kinv rows, X_j in registers),
// wave -> 16 rows i (wave-uniform: X_i, alpha_i come through scalar loads).  One wave reduction per
// parameter per workgroup at the end.
template <int DP>
__global__ __launch_bounds__(256) void nlml_grad_kernel(ModelDev m, const double* __restrict__ Kinv,
                                                        double* __restrict__ partial) {
  __shared__ double red[4][DP + 2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // G_ij dK_ij is symmetric: only the tiles on and below the diagonal are visited (K^-1 is produced for
  // those tiles only), off-diagonal ones count twice
  if (blockIdx.x > blockIdx.y) {
    if (tid < m.d + 2) partial[(int64_t)tid * gridDim.x * gridDim.y + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = 0.0;
    return;
  }
  const double sym = (blockIdx.x == blockIdx.y) ? 1.0 : 2.0;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;
  const int64_t i0 = (int64_t)blockIdx.y * 64 + w * 16;
  const int d = m.d;
  const bool jv = j < m.N;
  double xj[DP], acc[DP + 2];
#pragma unroll
  for (int c = 0; c < DP; ++c) xj[c] = m.Xs[j * DP + c];  // j < Npad always; padding rows are zero
#pragma unroll
  for (int c = 0; c < DP + 2; ++c) acc[c] = 0.0;
  const double aj = m.alpha[j];
  const cptr xs = as_const(m.Xs);
  const cptr al = as_const(m.alpha);
  for (int r = 0; r < 16; ++r) {
    const int64_t i = i0 + r;
    if (i >= m.N) break;  // wave-uniform
    const double G = jv ? Kinv[i * m.Npad + j] - al[i] * aj : 0.0;
    double r2 = 0.0, ds2[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      const double t = xs[i * DP + c] - xj[c];
      ds2[c] = t * t;
      r2 += ds2[c];
    }
    const double f1 = G * kernel_dr2(m.kind, r2, m.variance);  // variance * f'(r2)
    const double kij = kernel_rt(m.kind, r2, m.variance);      // without the noise
#pragma unroll
    for (int c = 0; c < DP; ++c) acc[c] = fma(f1, ds2[c], acc[c]);
    acc[DP] = fma(G, kij, acc[DP]);
    acc[DP + 1] += (i == j) ? G : 0.0;
  }
#pragma unroll
  for (int c = 0; c < DP + 2; ++c) {
    const double s = wave_sum(acc[c]);
    if (lane == 0) red[w][c] = s;
  }
  __syncthreads();
  if (tid < d + 2) {
    // slot order of `partial`: d lengthscale terms, variance, noise
    const int src = tid < d ? tid : DP + (tid - d);
    double v = sym * ((red[0][src] + red[1][src]) + (red[2][src] + red[3][src]));
    if (tid < d) v *= -2.0 / m.ls[tid];
    else if (tid == d) v /= m.variance;
    const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[(int64_t)tid * gridDim.x * gridDim.y + b] = v;  // [term][block]: the final reduction reads it coalesced
  }
}

// out[0] = nlml, out[1..d] = d/d lengthscale, out[d+1] = d/d variance, out[d+2] = d/d noise,
// out[d+3] = d/d mean.  One workgroup; fixed summation order.
// Batched value-only form (tgp_nlml_trial_batch): member blockIdx.x has its factor at L + x l_stride, its z (passed as
// both err and m.alpha) at + x v_stride, its result at out + x o_stride -- the same arithmetic in the same order.
__global__ __launch_bounds__(1024) void nlml_final_kernel(ModelDev m, const double* __restrict__ L,
                                                          const double* __restrict__ err,
                                                          const double* __restrict__ partial, int64_t nblocks,
                                                          double* __restrict__ out, int64_t l_stride = 0,
                                                          int64_t v_stride = 0, int64_t o_stride = 0) {
  L += (int64_t)blockIdx.x * l_stride;
  err += (int64_t)blockIdx.x * v_stride;
  m.alpha += (int64_t)blockIdx.x * v_stride;
  out += (int64_t)blockIdx.x * o_stride;
  // up to 1024 threads: the N diagonal entries of L are N separate cache lines and the block partials are d + 2 rows of
  // `nblocks` values -- with 256 threads and a block-major layout this kernel walked them 16 deep per thread
  // (100 us at N = 4096, as long as the N^2 pair reduction before it)
  __shared__ double red[16][NG_MAXP + 3];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int d = m.d, np = d + 2;
  double acc[NG_MAXP + 3];
  for (int c = 0; c < np + 3; ++c) acc[c] = 0.0;
  for (int c = 0; c < np; ++c)
    for (int64_t b = tid; b < nblocks; b += blockDim.x) acc[c] += partial[(int64_t)c * nblocks + b];
  for (int64_t i = tid; i < m.N; i += blockDim.x) {
    acc[np] += err[i] * m.alpha[i];
    acc[np + 1] += log(L[i * m.Npad + i]);
    acc[np + 2] += m.alpha[i];
  }
  for (int c = 0; c < np + 3; ++c) {
    const double s = wave_sum(acc[c]);
    if (lane == 0) red[w][c] = s;
  }
  __syncthreads();
  if (tid == 0) {
    auto tot = [&](int c) {
      double t = 0.0;
      for (int g = 0; g < (int)(blockDim.x >> 6); ++g) t += red[g][c];
      return t;
    };
    out[0] = 0.5 * tot(np) + tot(np + 1) + 0.5 * (double)m.N * 1.8378770664093453;  // log(2 pi)
    for (int c = 0; c < np; ++c) out[1 + c] = 0.5 * tot(c);
    out[1 + np] = -tot(np + 2);
  }
}

// value only (find_best_model_initialization compares losses, no gradient): no K^-1, no pair reduction
void launch_nlml_value(hipStream_t s, const ModelDev& m, const double* L, const double* err, double* out) {
  hipLaunchKernelGGL(nlml_final_kernel, dim3(1), dim3(m.N > 1024 ? 1024 : 256), 0, s, m, L, err, (const double*)nullptr, (int64_t)0, out);
}

// B members' values in one launch (value only: err = alpha = z)
void launch_nlml_value_batch(hipStream_t s, const ModelDev& m, const double* L, const double* z, double* out, int B,
                             int64_t l_stride, int64_t v_stride, int64_t o_stride) {
  ModelDev mm = m;
  mm.alpha = z;
  hipLaunchKernelGGL(nlml_final_kernel, dim3((unsigned)B), dim3(m.N > 1024 ? 1024 : 256), 0, s, mm, L, z, (const double*)nullptr,
                     (int64_t)0, out, l_stride, v_stride, o_stride);
}

int64_t nlml_blocks(int64_t Npad) { return (Npad / 64) * (Npad / 64); }

void launch_nlml(hipStream_t s, const ModelDev& m, const double* Kinv, const double* L, const double* err,
                 double* partial, double* out) {
  dim3 grid((unsigned)(m.Npad / 64), (unsigned)(m.Npad / 64));
  switch (m.dp) {
    case 2: hipLaunchKernelGGL(nlml_grad_kernel<2>, grid, dim3(256), 0, s, m, Kinv, partial); break;
    case 4: hipLaunchKernelGGL(nlml_grad_kernel<4>, grid, dim3(256), 0, s, m, Kinv, partial); break;
    case 6: hipLaunchKernelGGL(nlml_grad_kernel<6>, grid, dim3(256), 0, s, m, Kinv, partial); break;
    case 8: hipLaunchKernelGGL(nlml_grad_kernel<8>, grid, dim3(256), 0, s, m, Kinv, partial); break;
    case 16: hipLaunchKernelGGL(nlml_grad_kernel<16>, grid, dim3(256), 0, s, m, Kinv, partial); break;
    default: hipLaunchKernelGGL(nlml_grad_kernel<32>, grid, dim3(256), 0, s, m, Kinv, partial); break;
  }
  hipLaunchKernelGGL(nlml_final_kernel, dim3(1), dim3(m.N > 1024 ? 1024 : 256), 0, s, m, L, err, partial, nlml_blocks(m.Npad), out);
}

}  // namespace tgp
