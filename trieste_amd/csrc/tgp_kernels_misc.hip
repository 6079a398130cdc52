// Cold-path and tail kernels for gfx950: posterior mean only, (value, index) reductions, top-k
// passes, Philox candidate sampling, the batch Monte-Carlo EI tail and decoupled-trajectory
// evaluation.
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

namespace tgp {

// ---------------------------------------------------------------------------------------------
// mean[j] = sum_k k(x_j, X_k) alpha_k + c.  One thread per candidate, training rows via scalar
// loads.  == the mean half of GPflowPredictor.predict_encoded (interface.py:119-124); used for
// eta = min_i mean(X_i) (function.py:145-149) where no variance is needed.
template <int DP>
__global__ __launch_bounds__(256) void predict_mean_kernel(ModelDev m, const double* __restrict__ Xq,
                                                           int64_t M, double* __restrict__ mean) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = j < M;
  const int d = m.d;
  double xq[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) xq[c] = (c < d && valid) ? Xq[j * d + c] / as_const(m.ls)[c] : 0.0;
  double acc = 0.0;
  const cptr xs = as_const(m.Xs);
  const cptr al = as_const(m.alpha);
  for (int64_t k = 0; k < m.N; ++k) {
    double r2 = 0.0;
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      const double t = xq[c] - xs[k * DP + c];
      r2 = fma(t, t, r2);
    }
    acc = fma(kernel_rt(m.kind, r2, m.variance), al[k], acc);
  }
  if (valid) mean[j] = acc + m.mean_const;
}

// Few query points (eta: the N training inputs): one WAVE per query, lanes stride over the training
// rows, one wave reduction.  4096 queries fill 1024 SIMDs four times over, where the thread-per-
// query form above would occupy 16 CUs (2.1 ms -> tens of us at N = 4096).
template <int DP>
__global__ __launch_bounds__(256) void predict_mean_wave_kernel(ModelDev m, const double* __restrict__ Xq,
                                                                int64_t M, double* __restrict__ mean) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= M) return;  // wave-uniform
  const int d = m.d;
  double xq[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) xq[c] = (c < d) ? Xq[j * d + c] / as_const(m.ls)[c] : 0.0;
  double acc = 0.0;
  for (int64_t k = lane; k < m.N; k += 64) {
    double r2 = 0.0;
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      const double t = xq[c] - m.Xs[k * DP + c];
      r2 = fma(t, t, r2);
    }
    acc = fma(kernel_rt(m.kind, r2, m.variance), m.alpha[k], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) mean[j] = acc + m.mean_const;
}

void launch_predict_mean(hipStream_t s, const ModelDev& m, const double* Xq, int64_t M, double* mean) {
  if (M <= 32768) {
    dim3 g((unsigned)((M + 3) / 4)), b(256);
    switch (m.dp) {
      case 2: hipLaunchKernelGGL(predict_mean_wave_kernel<2>, g, b, 0, s, m, Xq, M, mean); break;
      case 4: hipLaunchKernelGGL(predict_mean_wave_kernel<4>, g, b, 0, s, m, Xq, M, mean); break;
      case 6: hipLaunchKernelGGL(predict_mean_wave_kernel<6>, g, b, 0, s, m, Xq, M, mean); break;
      case 8: hipLaunchKernelGGL(predict_mean_wave_kernel<8>, g, b, 0, s, m, Xq, M, mean); break;
      case 16: hipLaunchKernelGGL(predict_mean_wave_kernel<16>, g, b, 0, s, m, Xq, M, mean); break;
      default: hipLaunchKernelGGL(predict_mean_wave_kernel<32>, g, b, 0, s, m, Xq, M, mean); break;
    }
    return;
  }
  dim3 g((unsigned)((M + 255) / 256)), b(256);
  switch (m.dp) {
    case 2: hipLaunchKernelGGL(predict_mean_kernel<2>, g, b, 0, s, m, Xq, M, mean); break;
    case 4: hipLaunchKernelGGL(predict_mean_kernel<4>, g, b, 0, s, m, Xq, M, mean); break;
    case 6: hipLaunchKernelGGL(predict_mean_kernel<6>, g, b, 0, s, m, Xq, M, mean); break;
    case 8: hipLaunchKernelGGL(predict_mean_kernel<8>, g, b, 0, s, m, Xq, M, mean); break;
    case 16: hipLaunchKernelGGL(predict_mean_kernel<16>, g, b, 0, s, m, Xq, M, mean); break;
    default: hipLaunchKernelGGL(predict_mean_kernel<32>, g, b, 0, s, m, Xq, M, mean); break;
  }
}

// ---------------------------------------------------------------------------------------------
// Second half of a row-group-split sweep (tgp_kernels_sweep.inc, SPLIT): sum the groups' partial
// (k*.alpha, sum c^2) in a fixed order, then the same tail as the fused epilogue -- clip, acquisition value,
// outputs, per-block (max value, min index).  One 128-thread workgroup per candidate block.
__global__ __launch_bounds__(128) void sweep_combine_kernel(SweepArgs a) {
  __shared__ double bvs[2];
  __shared__ int64_t bis[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t M = a.M_dev ? *a.M_dev : a.M;  // repair pass: the count of flagged candidates lives on the device
  const int64_t nblk = (M + 127) / 128;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t cj = blk * 128 + tid;
    double val = -INFINITY;
    int64_t gidx = INT64_MAX;
    if (cj < M) {
      double m = 0.0, sq = 0.0;
      for (int g = 0; g < a.split_g; ++g) {
        const double* pp = a.part + ((size_t)blk * a.split_g + g) * 256;
        m += pp[tid];
        sq += pp[128 + tid];
      }
      const double mean = m + a.m.mean_const;
      const double var = fmax(a.m.variance - sq, VAR_FLOOR);
      if (a.mean_out) a.mean_out[cj] = mean;
      if (a.var_out) a.var_out[cj] = var;
      if (a.acq_kind >= 0) {
        const double v = acq_tail(a.acq_kind, a.acq_param, mean, var, a.m.noise);
        if (a.acq_out) a.acq_out[cj] = v;
        if (!(v != v)) {
          val = v;
          gidx = a.index_base + cj;
        }
      }
    }
    if (a.blk_val) {
      wave_argmax(val, gidx);
      if (lane == 0) {
        bvs[w] = val;
        bis[w] = gidx;
      }
      __syncthreads();
      if (tid == 0) {
        double v0 = bvs[0];
        int64_t i0 = bis[0];
        if (better(bvs[1], bis[1], v0, i0)) {
          v0 = bvs[1];
          i0 = bis[1];
        }
        a.blk_val[blk] = v0;
        a.blk_idx[blk] = i0;
      }
      __syncthreads();
    }
  }
}

void launch_sweep_combine(hipStream_t s, const SweepArgs& a, int64_t nblk) {
  hipLaunchKernelGGL(sweep_combine_kernel, dim3((unsigned)nblk), dim3(128), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// final (max value, min index) over per-workgroup partials: one workgroup.
__global__ __launch_bounds__(256) void argmax_final_kernel(const double* __restrict__ bv,
                                                           const int64_t* __restrict__ bi, int64_t n,
                                                           double* out_val, int64_t* out_idx) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  double v = -INFINITY;
  int64_t i = INT64_MAX;
  for (int64_t t = threadIdx.x; t < n; t += 256) {
    if (better(bv[t], bi[t], v, i)) {
      v = bv[t];
      i = bi[t];
    }
  }
  wave_argmax(v, i);
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (better(sv[w], si[w], v, i)) {
        v = sv[w];
        i = si[w];
      }
    *out_val = v;
    *out_idx = i;
  }
}
void launch_argmax_final(hipStream_t s, const double* blk_val, const int64_t* blk_idx, int64_t n,
                         double* out_val, int64_t* out_idx) {
  hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, blk_val, blk_idx, n, out_val,
                     out_idx);
}

__global__ __launch_bounds__(256) void min_value_kernel(const double* __restrict__ v, int64_t n,
                                                        double* out) {
  __shared__ double sv[4];
  double m = INFINITY;
  for (int64_t t = threadIdx.x; t < n; t += 256) m = fmin(m, v[t]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmin(fmin(sv[0], sv[1]), fmin(sv[2], sv[3]));
}
void launch_min_value(hipStream_t s, const double* v, int64_t n, double* out) {
  hipLaunchKernelGGL(min_value_kernel, dim3(1), dim3(256), 0, s, v, n, out);
}

// ---------------------------------------------------------------------------------------------
// One top-k pass: best element strictly after (pv, pi) in the order (value desc, index asc).
// == one extraction step of tf.math.top_k in generate_initial_points (optimizer.py:326-335).
// The threshold (previous winner) lives on the device (out_val/out_idx[t-1]): the k passes are enqueued
// back to back and the host reads all k results once.
constexpr int TOPK_BLOCKS = 512;
__global__ __launch_bounds__(256) void topk_pass_kernel(const double* __restrict__ vals, int64_t M,
                                                        int64_t index_base, const double* __restrict__ prev_val,
                                                        const int64_t* __restrict__ prev_idx,
                                                        double* __restrict__ sv, int64_t* __restrict__ si) {
  __shared__ double wv[4];
  __shared__ int64_t wi[4];
  const bool first = prev_val == nullptr;
  const double pv = first ? 0.0 : *prev_val;
  const int64_t pi = first ? 0 : *prev_idx;
  double v = -INFINITY;
  int64_t i = INT64_MAX;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < M; t += (int64_t)TOPK_BLOCKS * 256) {
    const double x = vals[t];
    const int64_t xi = index_base + t;
    if (x != x) continue;
    const bool after = first || (x < pv) || (x == pv && xi > pi);
    if (after && better(x, xi, v, i)) {
      v = x;
      i = xi;
    }
  }
  wave_argmax(v, i);
  if ((threadIdx.x & 63) == 0) {
    wv[threadIdx.x >> 6] = v;
    wi[threadIdx.x >> 6] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (better(wv[w], wi[w], v, i)) {
        v = wv[w];
        i = wi[w];
      }
    sv[blockIdx.x] = v;
    si[blockIdx.x] = i;
  }
}
void launch_topk_pass(hipStream_t s, const double* vals, int64_t M, int64_t index_base, const double* prev_val,
                      const int64_t* prev_idx, double* scratch_val, int64_t* scratch_idx, double* out_val,
                      int64_t* out_idx) {
  hipLaunchKernelGGL(topk_pass_kernel, dim3(TOPK_BLOCKS), dim3(256), 0, s, vals, M, index_base, prev_val,
                     prev_idx, scratch_val, scratch_idx);
  launch_argmax_final(s, scratch_val, scratch_idx, TOPK_BLOCKS, out_val, out_idx);
}

// Small candidate sets (EGO's default initial sweep: max(5000, 1000 d) values) are sorted instead of scanned k
// times.
constexpr int64_t TOPK_SMALL_MAX = 65536;
// A workgroup holds 8192 (value, offset) pairs in LDS (96 KiB) and runs a bitonic
// network in the order (value descending, index ascending) -- 91 compare-exchange stages instead of k passes
// over the values.  NaNs rank after everything (they are never
// "better"); slots beyond the non-NaN count come out as (-inf, INT64_MAX), like in the pass form.  More than 8192
// values: one workgroup per 8192-chunk emits its top k, a second launch sorts the <= 8 k survivors.
constexpr int TOPK_SORT_N = 8192;
__global__ __launch_bounds__(1024) void topk_sort_kernel(const double* __restrict__ vals,
                                                         const int64_t* __restrict__ idxs, int64_t n,
                                                         int64_t index_base, int k, double* __restrict__ out_val,
                                                         int64_t* __restrict__ out_idx) {
  __shared__ double key[TOPK_SORT_N];
  __shared__ int off[TOPK_SORT_N];
  const int tid = threadIdx.x;
  const int64_t first = (int64_t)blockIdx.x * TOPK_SORT_N;
  for (int e = tid; e < TOPK_SORT_N; e += 1024) {
    double x = -INFINITY;
    int o = INT32_MAX;
    if (first + e < n) {
      const double v = vals[first + e];
      const int64_t gi = idxs ? idxs[first + e] : index_base + first + e;
      if (v == v && gi != INT64_MAX) {
        x = v;
        o = (int)(gi - index_base);
      }
    }
    key[e] = x;
    off[e] = o;
  }
  __syncthreads();
  for (int size = 2; size <= TOPK_SORT_N; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < TOPK_SORT_N / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));  // lower index of the pair
        const int hi = lo + stride;
        const bool first_goes_low = (lo & size) == 0;  // direction of this bitonic block
        const double a = key[lo], b = key[hi];
        const int ia = off[lo], ib = off[hi];
        const bool b_before_a = (b > a) || (b == a && ib < ia);
        if (b_before_a == first_goes_low) {
          key[lo] = b;
          key[hi] = a;
          off[lo] = ib;
          off[hi] = ia;
        }
      }
      __syncthreads();
    }
  }
  for (int t = tid; t < k; t += 1024) {
    const int o = off[t];
    out_val[(int64_t)blockIdx.x * k + t] = key[t];
    out_idx[(int64_t)blockIdx.x * k + t] = o == INT32_MAX ? INT64_MAX : index_base + o;
  }
}

// scratch_val / scratch_idx: at least 8 * k entries (used when M > 8192)
void launch_topk_small(hipStream_t s, const double* vals, int64_t M, int64_t index_base, int k, double* out_val,
                       int64_t* out_idx, double* scratch_val, int64_t* scratch_idx) {
  const int nb = (int)((M + TOPK_SORT_N - 1) / TOPK_SORT_N);
  if (nb == 1) {
    hipLaunchKernelGGL(topk_sort_kernel, dim3(1), dim3(1024), 0, s, vals, (const int64_t*)nullptr, M, index_base, k,
                       out_val, out_idx);
    return;
  }
  hipLaunchKernelGGL(topk_sort_kernel, dim3(nb), dim3(1024), 0, s, vals, (const int64_t*)nullptr, M, index_base, k,
                     scratch_val, scratch_idx);
  hipLaunchKernelGGL(topk_sort_kernel, dim3(1), dim3(1024), 0, s, (const double*)scratch_val,
                     (const int64_t*)scratch_idx, (int64_t)nb * k, index_base, k, out_val, out_idx);
}
int64_t topk_small_max() { return TOPK_SMALL_MAX; }

// ---------------------------------------------------------------------------------------------
// Local penalization (reference acquisition/function/greedy_batch.py: PenalizedAcquisition 250-269,
// soft_local_penalizer.__call__ 341-354, hard_local_penalizer.__call__ 376-389): the acquisition values of a
// sweep are multiplied by prod_p phi_p(x), phi_p a function of |x - pending_p| (plain Euclidean norm of the
// UNSCALED inputs), radius_p and scale_p.  One thread per candidate, the <= 1024 pending points through
// wave-uniform scalar loads: M (d + 2) 8 B of HBM traffic against the sweep's N^2 flops per candidate.
//   soft: phi = Phi((dist - r) / s);   hard: phi = ((dist / (r + s))^-5 + 1)^(-1/5)
__device__ __forceinline__ double penalty_factor(int kind, double dist, double r, double sc) {
  if (kind == 1) return normal_cdf((dist - r) / sc);
  return pow(pow(dist / (r + sc), -5.0) + 1.0, -0.2);
}
// d phi / d dist
__device__ __forceinline__ double penalty_slope(int kind, double dist, double r, double sc) {
  if (kind == 1) return normal_pdf((dist - r) / sc) / sc;
  const double u = dist / (r + sc);
  // d/du (u^-5 + 1)^(-1/5) = u^-6 (u^-5 + 1)^(-6/5)
  return pow(u, -6.0) * pow(pow(u, -5.0) + 1.0, -1.2) / (r + sc);
}

__global__ __launch_bounds__(256) void penalize_kernel(double* __restrict__ vals, const double* __restrict__ Xq,
                                                       int64_t M, int d, int kind, int P, int init_one,
                                                       const double* __restrict__ pend,
                                                       const double* __restrict__ radius,
                                                       const double* __restrict__ scale) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double x[MAX_D];
  for (int c = 0; c < d; ++c) x[c] = Xq[m * d + c];
  double prod = 1.0;
  for (int p = 0; p < P; ++p) {
    double r2 = 0.0;
    for (int c = 0; c < d; ++c) {
      const double t = x[c] - pend[(int64_t)p * d + c];
      r2 += t * t;
    }
    prod *= penalty_factor(kind, sqrt(r2), radius[p], scale[p]);
  }
  // the reference multiplies in log space, exp(log a + log phi) (greedy_batch.py:266-269): identical to the
  // product for a, phi > 0; a = 0 or phi = 0 give exp(-inf) = 0 there and 0 here
  vals[m] = init_one ? prod : vals[m] * prod;
}

// value and gradient of the penalised acquisition at P' points: (a phi)' = phi a' + a phi',
// phi' = sum_p slope_p (x - x_p) / dist_p prod_{q != p} phi_q  (O(P^2) per point; P' is a few hundred).
__global__ __launch_bounds__(64) void penalize_grad_kernel(double* __restrict__ val, double* __restrict__ grad,
                                                           const double* __restrict__ Xq, int64_t Pq, int d,
                                                           int kind, int P, const double* __restrict__ pend,
                                                           const double* __restrict__ radius,
                                                           const double* __restrict__ scale) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Pq) return;
  double x[MAX_D], gphi[MAX_D];
  for (int c = 0; c < d; ++c) {
    x[c] = Xq[m * d + c];
    gphi[c] = 0.0;
  }
  double prod = 1.0;
  for (int p = 0; p < P; ++p) {
    double r2 = 0.0;
    for (int c = 0; c < d; ++c) {
      const double t = x[c] - pend[(int64_t)p * d + c];
      r2 += t * t;
    }
    const double dist = sqrt(r2);
    const double f = penalty_factor(kind, dist, radius[p], scale[p]);
    prod *= f;
    double others = 1.0;  // prod_{q != p} phi_q
    for (int q = 0; q < P; ++q) {
      if (q == p) continue;
      double s2 = 0.0;
      for (int c = 0; c < d; ++c) {
        const double t = x[c] - pend[(int64_t)q * d + c];
        s2 += t * t;
      }
      others *= penalty_factor(kind, sqrt(s2), radius[q], scale[q]);
    }
    // at dist = 0 the norm has no gradient (the reference's autodiff returns NaN there); 0 keeps L-BFGS-B alive
    const double w = dist > 0.0 ? penalty_slope(kind, dist, radius[p], scale[p]) * others / dist : 0.0;
    for (int c = 0; c < d; ++c) gphi[c] += w * (x[c] - pend[(int64_t)p * d + c]);
  }
  const double a = val[m];
  for (int c = 0; c < d; ++c) grad[m * d + c] = prod * grad[m * d + c] + a * gphi[c];
  val[m] = a * prod;
}

void launch_penalize(hipStream_t s, double* vals, const double* Xq, int64_t M, int d, int kind, int P,
                     const double* pend, const double* radius, const double* scale, bool init_one) {
  if (M <= 0 || P <= 0) return;
  hipLaunchKernelGGL(penalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, vals, Xq, M, d, kind, P,
                     init_one ? 1 : 0, pend, radius, scale);
}
void launch_penalize_grad(hipStream_t s, double* val, double* grad, const double* Xq, int64_t Pq, int d, int kind,
                          int P, const double* pend, const double* radius, const double* scale) {
  if (Pq <= 0 || P <= 0) return;
  hipLaunchKernelGGL(penalize_grad_kernel, dim3((unsigned)((Pq + 63) / 64)), dim3(64), 0, s, val, grad, Xq, Pq, d,
                     kind, P, pend, radius, scale);
}

// ---------------------------------------------------------------------------------------------
// Entropy-search acquisition values from a sweep's (mean, var) arrays (entropy_tail in tgp_dev.hpp): one thread
// per candidate, the S min-value samples through scalar loads.  var_twin != nullptr adds GIBBON's repulsion term
//   weight / 2 * (log(var_twin + noise) - log(var + noise)),
// var_twin being the variance of the model conditioned additionally on the pending points:
// yvar - A^T (B + noise I)^-1 A of gibbon_repulsion_term.__call__ (entropy.py:596-612) IS that variance + noise.
__global__ __launch_bounds__(256) void entropy_tail_kernel(const double* __restrict__ mean,
                                                           const double* __restrict__ var,
                                                           const double* __restrict__ var_twin, int64_t M, int acq,
                                                           double noise, const double* __restrict__ samples, int S,
                                                           double weight, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  double v, dmu, dvar;
  entropy_tail(acq, mean[i], var[i], noise, samples, S, v, dmu, dvar);
  if (var_twin) v += 0.5 * weight * (log(var_twin[i] + noise) - log(var[i] + noise));
  out[i] = v;
}

void launch_entropy_tail(hipStream_t s, const double* mean, const double* var, const double* var_twin, int64_t M,
                         int acq, double noise, const double* samples, int S, double weight, double* out) {
  if (M <= 0) return;
  hipLaunchKernelGGL(entropy_tail_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, mean, var, var_twin, M,
                     acq, noise, samples, S, weight, out);
}

// ---------------------------------------------------------------------------------------------
// GIBBON's repulsion through a rank-m update instead of a second sweep.  The twin is this model plus m appended
// rows, so its factor inverse W' agrees with W on the first N rows and
//     var'(x) = k** - sum_{r < N + m} (W'_r . k'(x))^2 = var(x) - sum_{r = N}^{N + m - 1} (W'_r . k'(x))^2:
// m dot products of length N + m per candidate (launch_kernel_sums) instead of (N + m)^2 / 2 flops.
__global__ __launch_bounds__(256) void lowrank_var_kernel(const double* __restrict__ var,
                                                          const double* __restrict__ u, int64_t M, int m,
                                                          double* __restrict__ var_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  double s = 0.0;
  for (int b = 0; b < m; ++b) s = fma(u[i * m + b], u[i * m + b], s);
  var_out[i] = fmax(var[i] - s, VAR_FLOOR);
}
void launch_lowrank_var(hipStream_t s, const double* var, const double* u, int64_t M, int m, double* var_out) {
  if (M <= 0) return;
  hipLaunchKernelGGL(lowrank_var_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, var, u, M, m, var_out);
}

// out[k][b] = W[(row0 + b) * ld + k], k < n: the last m rows of the twin's W as [n][m] weight columns
__global__ void rows_to_columns_kernel(const double* __restrict__ W, int64_t ld, int64_t row0, int m, int64_t n,
                                       double* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  for (int b = 0; b < m; ++b) out[k * m + b] = W[(row0 + b) * ld + k];
}
void launch_rows_to_columns(hipStream_t s, const double* W, int64_t ld, int64_t row0, int m, int64_t n, double* out) {
  hipLaunchKernelGGL(rows_to_columns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ld, row0, m, n, out);
}

// *flag = 1 if a[i] != b[i] for some i < n (flag must be zeroed by the caller)
__global__ void prefix_differs_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                      int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) *flag = 1;
}
void launch_prefix_differs(hipStream_t s, const double* a, const double* b, int64_t n, int* flag) {
  if (n <= 0) return;
  hipLaunchKernelGGL(prefix_differs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, n, flag);
}

// ---------------------------------------------------------------------------------------------
// Box.sample (reference space.py:843-867) on device: uniform in [lower, upper).
// Three separately rounded operations: the numpy restatement oracle/philox.py is bit-exact (hipcc's default
// -ffp-contract=fast would fuse the multiply and the add into one fma, which rounds differently).
__device__ __forceinline__ double box_map(double u, double lo, double up) {
#pragma clang fp contract(off)
  const double width = up - lo;
  const double scaled = width * u;
  return lo + scaled;
}
__global__ void sample_box_kernel(uint64_t seed, int64_t first, int64_t M, int d,
                                  const double* __restrict__ lower, const double* __restrict__ upper,
                                  double* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * d) return;
  const int64_t row = t / d;
  const int c = (int)(t % d);
  const double u = philox_uniform(seed, (uint64_t)((first + row) * d + c));
  out[t] = box_map(u, lower[c], upper[c]);
}
void launch_sample_box(hipStream_t s, uint64_t seed, int64_t first, int64_t M, int d,
                       const double* lower, const double* upper, double* out) {
  const int64_t n = M * d;
  hipLaunchKernelGGL(sample_box_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, first,
                     M, d, lower, upper, out);
}

// ---------------------------------------------------------------------------------------------
// Batch Monte-Carlo EI tail, ONE WAVE per group of q points (a 64-thread workgroup: no s_barrier anywhere):
//   Lq = chol(cov + jitter I);  samples_s = mean + Lq eps[:, s];  out = mean_s max(eta - min_j, 0)
// == BatchReparametrizationSampler.sample (sampler.py:278-287) + batch_monte_carlo_expected_
// improvement.__call__ (function.py:1183-1186).  (SURVEY K2 batched, K8.)
// Round 2's form (one 256-thread workgroup per group, three barriers per pivot, eps re-read from L2 for every (j, k))
// took 33 ms for C4's 10^5 groups, 7 % of the step; its arithmetic is kept operation for operation:
//  * factorisation: lane i owns row i, kept in its own LDS row (odd row stride: conflict-free); the pivot and the
//    multiplier l_kj reach every lane through v_readlane (-> a scalar operand of the FMA), so no lane ever reads
//    another lane's LDS data: no barrier, no fence, and the compiler is free to pipeline the row updates.  Entries
//    right of the diagonal are updated along with the rest (never read);
//  * samples: lane = sample; the q base draws of a sample sit in registers (loops unrolled to the template bound QP,
//    left by wave-uniform branches), L_jk comes from LDS as a broadcast read (one read + one FMA per term, q(q+1)/2
//    terms), RG rows at a time: independent FMA chains hide the f64 latency.
template <int QP>
__global__ __launch_bounds__(64, 2) void qei_tail_kernel(const double* __restrict__ mean,
                                                         const double* __restrict__ cov, int64_t G, int q,
                                                         const double* __restrict__ eps, int S, double eta,
                                                         double jitter, double* __restrict__ out,
                                                         double* __restrict__ samples_out,
                                                         int* __restrict__ info) {
  extern __shared__ double qei_lds[];
  const int64_t g = blockIdx.x;
  const int lane = threadIdx.x;
  const int ldq = q | 1;  // odd row stride
  double* const Ls = qei_lds;       // [q][ldq] (+ QP of slack: the last row group reads, and drops, rows beyond q)
  double* const mu = qei_lds + q * ldq + QP;
  auto bcast = [](double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
  };
  const bool live = lane < q;
  double* const Lrow = Ls + (live ? lane : 0) * ldq;
  if (live) {
    const double* const crow = cov + (g * q + lane) * q;
    for (int k = 0; k < q; ++k) Lrow[k] = crow[k] + (k == lane ? jitter : 0.0);
    mu[lane] = mean[g * q + lane];
  }
  for (int j = 0; j < q; ++j) {
    const double x = Lrow[j];
    double dj = bcast(x, j);
    if (!(dj > 0.0)) {
      if (lane == 0) atomicCAS(info, 0, (int)(g % 2000000000) + 1);
      dj = 1.0;
    }
    const double sd = sqrt(dj);
    const double lij = lane == j ? sd : x / sd;
    const bool below = live && lane > j;
    if (live && lane >= j) Lrow[j] = lij;
    const double nl = -lij;
    for (int k = j + 1; k < q; ++k) {
      const double lkj = bcast(lij, k);
      if (below) Lrow[k] = fma(nl, lkj, Lrow[k]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // from here on every lane reads every row
  constexpr int RG = 2;
  double acc = 0.0;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool valid = s < S;
    double er[QP];
#pragma unroll
    for (int k = 0; k < QP; ++k) er[k] = (k < q && valid) ? eps[(int64_t)k * S + s] : 0.0;
    double mn = INFINITY;
#pragma unroll
    for (int j0 = 0; j0 < QP; j0 += RG) {
      if (j0 < q) {  // (wave-uniform)
        double v[RG];
        const double* Lj[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
          const int j = j0 + r < q ? j0 + r : q - 1;
          Lj[r] = Ls + j * ldq;
          v[r] = mu[j];
        }
#pragma unroll
        for (int k = 0; k < j0 + RG; ++k)
#pragma unroll
          for (int r = 0; r < RG; ++r)
            if (k <= j0 + r) v[r] = fma(Lj[r][k], er[k], v[r]);
#pragma unroll
        for (int r = 0; r < RG; ++r)
          if (j0 + r < q) {
            if (samples_out && valid) samples_out[(g * S + s) * q + j0 + r] = v[r];  // [G][S][q]
            mn = fmin(mn, v[r]);
          }
      }
    }
    if (valid) acc += fmax(eta - mn, 0.0);
  }
  acc = wave_sum(acc);
  if (lane == 0 && out) out[g] = acc / (double)S;
}
template <int QP>
static void launch_qei_tail_qp(hipStream_t s, const double* mean, const double* cov, int64_t G, int q,
                               const double* eps, int S, double eta, double jitter, double* out,
                               double* samples_out, int* info) {
  const size_t lds = (size_t)(q * (q | 1) + QP + q) * sizeof(double);
  hipLaunchKernelGGL(qei_tail_kernel<QP>, dim3((unsigned)G), dim3(64), lds, s, mean, cov, G, q, eps, S, eta, jitter,
                     out, samples_out, info);
}
void launch_qei_tail(hipStream_t s, const double* mean, const double* cov, int64_t G, int q,
                     const double* eps, int S, double eta, double jitter, double* out, double* samples_out,
                     int* info) {
  if (q <= 8) launch_qei_tail_qp<8>(s, mean, cov, G, q, eps, S, eta, jitter, out, samples_out, info);
  else if (q <= 16) launch_qei_tail_qp<16>(s, mean, cov, G, q, eps, S, eta, jitter, out, samples_out, info);
  else if (q <= 32) launch_qei_tail_qp<32>(s, mean, cov, G, q, eps, S, eta, jitter, out, samples_out, info);
  else launch_qei_tail_qp<MAX_Q>(s, mean, cov, G, q, eps, S, eta, jitter, out, samples_out, info);
}

// ---------------------------------------------------------------------------------------------
// qEI AND its adjoints w.r.t. (mean, cov), ONE WAVE per group (round 6; the gradient of BatchMonteCarloExpectedImprovement for the
// L-BFGS-B refinement of a joint batch: reference optimizer.py:628-629 through sampler.py:276-287 and function.py:1183-1186):
//   L = chol(cov + jitter I);  f_s = mean + L eps[:, s];  value = mean_s max(eta - min_i f_s,i, 0)
//   d value / d f_s,i = -1/S  where i is the (first) arg-min of an improving sample  ->  gmean_i = sum_s ...,  Lbar = tril(sum_s df_s eps_s^T)
//   gcov = L^-T sym(Phi(L^T Lbar)) L^-1   (the Cholesky adjoint; Phi: lower triangle, diagonal halved), a clipped diagonal entry of
//   cov (interface.py:129-131: tf.clip_by_value) gets zero.
// Factorisation and samples as in qei_tail_kernel (same arithmetic, operation for operation: the value is tgp_qei's); every later
// phase gives lane c column c (or row r) of the q x q work matrix in LDS -- fixed summation orders, no atomics: bit-identical
// run to run.  q x q arithmetic per group, 10 ... 300 groups per call: none of it is a roofline item.
template <int QP>
__global__ __launch_bounds__(64) void qei_grad_tail_kernel(const double* __restrict__ mean, const double* __restrict__ cov,
                                                           int64_t G, int q, const double* __restrict__ eps, int S, double eta,
                                                           double jitter, double* __restrict__ val,
                                                           double* __restrict__ gmean, double* __restrict__ gcov,
                                                           int* __restrict__ info) {
  extern __shared__ double qei_lds[];
  const int64_t g = blockIdx.x;
  const int lane = threadIdx.x;
  const int ldq = q | 1;  // odd row stride
  double* const Ls = qei_lds;                    // [q][ldq] (+ QP of slack, as in qei_tail_kernel)
  double* const Ms = Ls + q * ldq + QP;          // [q][ldq]: Lbar -> Q -> R -> L^-T R -> gcov, in place
  double* const mu = Ms + q * ldq;               // [QP]
  int* const jm = (int*)(mu + QP);               // [S]: arg-min row of an improving sample, -1 otherwise
  auto bcast = [](double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
  };
  const bool live = lane < q;
  double* const Lrow = Ls + (live ? lane : 0) * ldq;
  bool clipped = false;
  if (live) {
    const double* const crow = cov + (g * q + lane) * q;
    for (int k = 0; k < q; ++k) Lrow[k] = crow[k] + (k == lane ? jitter : 0.0);
    clipped = !(crow[lane] > VAR_FLOOR);
    mu[lane] = mean[g * q + lane];
    for (int i = 0; i < q; ++i) Ms[i * ldq + lane] = 0.0;
  }
  for (int j = 0; j < q; ++j) {
    const double x = Lrow[j];
    double dj = bcast(x, j);
    if (!(dj > 0.0)) {
      if (lane == 0) atomicCAS(info, 0, (int)(g % 2000000000) + 1);
      dj = 1.0;
    }
    const double sd = sqrt(dj);
    const double lij = lane == j ? sd : x / sd;
    const bool below = live && lane > j;
    if (live && lane >= j) Lrow[j] = lij;
    const double nl = -lij;
    for (int k = j + 1; k < q; ++k) {
      const double lkj = bcast(lij, k);
      if (below) Lrow[k] = fma(nl, lkj, Lrow[k]);
    }
  }
  __syncthreads();
  constexpr int RG = 2;
  double acc = 0.0;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool valid = s < S;
    double er[QP];
#pragma unroll
    for (int k = 0; k < QP; ++k) er[k] = (k < q && valid) ? eps[(int64_t)k * S + s] : 0.0;
    double mn = INFINITY;
    int jmin = 0;
#pragma unroll
    for (int j0 = 0; j0 < QP; j0 += RG) {
      if (j0 < q) {  // (wave-uniform)
        double v[RG];
        const double* Lj[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
          const int j = j0 + r < q ? j0 + r : q - 1;
          Lj[r] = Ls + j * ldq;
          v[r] = mu[j];
        }
#pragma unroll
        for (int k = 0; k < j0 + RG; ++k)
#pragma unroll
          for (int r = 0; r < RG; ++r)
            if (k <= j0 + r) v[r] = fma(Lj[r][k], er[k], v[r]);
#pragma unroll
        for (int r = 0; r < RG; ++r)
          if (j0 + r < q && v[r] < mn) {  // strict: the first index on ties
            mn = v[r];
            jmin = j0 + r;
          }
      }
    }
    const double imp = eta - mn;
    if (valid) {
      acc += fmax(imp, 0.0);
      jm[s] = imp > 0.0 ? jmin : -1;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) val[g] = acc / (double)S;
  __syncthreads();
  // Lbar (lower triangle) and gmean: lane c owns column c; the samples in order
  const double ninv = -1.0 / (double)S;
  if (live) {
    int cnt = 0;
    const double* const ec = eps + (int64_t)lane * S;
    for (int s = 0; s < S; ++s) {
      const int i = jm[s];  // (wave-uniform)
      if (i < 0) continue;
      if (i == lane) ++cnt;
      if (i >= lane) Ms[i * ldq + lane] += ec[s];
    }
    gmean[g * q + lane] = ninv * (double)cnt;
    // Q = tril(L^T Lbar) in place, column c: Q[a][c] = sum_{k >= a} L[k][a] Lbar[k][c]  (rows ascending: row a is read before it
    // is written and never again), then R's lower triangle = Q / 2 (Phi halves the diagonal, sym the rest)
    for (int a = lane; a < q; ++a) {
      double t = 0.0;
      for (int k = a; k < q; ++k) t = fma(Ls[k * ldq + a], ninv * Ms[k * ldq + lane], t);
      Ms[a * ldq + lane] = 0.5 * t;
    }
  }
  __syncthreads();
  if (live)
    for (int a = 0; a < lane; ++a) Ms[a * ldq + lane] = Ms[lane * ldq + a];  // R = R^T
  __syncthreads();
  if (live)  // Y = L^-T R: back substitution down column c
    for (int a = q - 1; a >= 0; --a) {
      double t = Ms[a * ldq + lane];
      for (int k = a + 1; k < q; ++k) t = fma(-Ls[k * ldq + a], Ms[k * ldq + lane], t);
      Ms[a * ldq + lane] = t / Ls[a * ldq + a];
    }
  __syncthreads();
  if (live) {  // gcov = Y L^-1: back substitution along row r
    double* const row = Ms + lane * ldq;
    for (int a = q - 1; a >= 0; --a) {
      double t = row[a];
      for (int k = a + 1; k < q; ++k) t = fma(-Ls[k * ldq + a], row[k], t);
      row[a] = t / Ls[a * ldq + a];
    }
    if (clipped) row[lane] = 0.0;
  }
  __syncthreads();
  if (live)
    for (int r = 0; r < q; ++r) gcov[(g * q + r) * q + lane] = Ms[r * ldq + lane];
}
template <int QP>
static void launch_qei_grad_tail_qp(hipStream_t s, const double* mean, const double* cov, int64_t G, int q, const double* eps,
                                    int S, double eta, double jitter, double* val, double* gmean, double* gcov, int* info) {
  const size_t lds = (size_t)(2 * q * (q | 1) + 2 * QP) * sizeof(double) + (size_t)((S + 1) & ~1) * sizeof(int);
  (void)hipFuncSetAttribute((const void*)qei_grad_tail_kernel<QP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(qei_grad_tail_kernel<QP>, dim3((unsigned)G), dim3(64), lds, s, mean, cov, G, q, eps, S, eta, jitter, val,
                     gmean, gcov, info);
}
size_t qei_grad_tail_lds_bytes(int q, int S) {
  return (size_t)(2 * q * (q | 1) + 2 * MAX_Q) * sizeof(double) + (size_t)((S + 1) & ~1) * sizeof(int);
}
void launch_qei_grad_tail(hipStream_t s, const double* mean, const double* cov, int64_t G, int q, const double* eps, int S,
                          double eta, double jitter, double* val, double* gmean, double* gcov, int* info) {
  if (q <= 8) launch_qei_grad_tail_qp<8>(s, mean, cov, G, q, eps, S, eta, jitter, val, gmean, gcov, info);
  else if (q <= 16) launch_qei_grad_tail_qp<16>(s, mean, cov, G, q, eps, S, eta, jitter, val, gmean, gcov, info);
  else if (q <= 32) launch_qei_grad_tail_qp<32>(s, mean, cov, G, q, eps, S, eta, jitter, val, gmean, gcov, info);
  else launch_qei_grad_tail_qp<MAX_Q>(s, mean, cov, G, q, eps, S, eta, jitter, val, gmean, gcov, info);
}

// final arg-min over per-workgroup partials for B trajectories: one workgroup per trajectory.
__global__ __launch_bounds__(256) void argmin_final_multi_kernel(const double* __restrict__ bv,
                                                                 const int64_t* __restrict__ bi,
                                                                 int64_t nblk, int B, double* out_val,
                                                                 int64_t* out_idx) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  const int b = blockIdx.x;
  double v = -INFINITY;
  int64_t i = INT64_MAX;
  for (int64_t t = threadIdx.x; t < nblk; t += 256) {
    const double x = -bv[t * B + b];
    const int64_t xi = bi[t * B + b];
    if (better(x, xi, v, i)) {
      v = x;
      i = xi;
    }
  }
  wave_argmax(v, i);
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (better(sv[w], si[w], v, i)) {
        v = sv[w];
        i = si[w];
      }
    out_val[b] = -v;
    out_idx[b] = i;
  }
}
void launch_argmin_final_multi(hipStream_t s, const double* blk_val, const int64_t* blk_idx,
                               int64_t nblk, int B, double* out_val, int64_t* out_idx) {
  hipLaunchKernelGGL(argmin_final_multi_kernel, dim3((unsigned)B), dim3(256), 0, s, blk_val, blk_idx,
                     nblk, B, out_val, out_idx);
}

// ---------------------------------------------------------------------------------------------
// Digit planes of W = L^-1 for the split-precision sweep (tgp_kernels_sweep_i8.inc): row i is scaled by
// S_i = 2 max_k |W_ik| (so |W_ik / S_i| <= 1/2), x = rint(W_ik / S_i * 2^(8 NS - 1)), and its NS (4 or 5) balanced
// base-256 digits (each in [-128, 127], most significant first) go to the planes
//   Wq[s][k / 32][i][k % 32]      (a [256 rows x 32 k] operand tile of one plane is 8 KiB contiguous).
// One workgroup per 32 rows; thread (r = tid >> 3, c = tid & 7) owns 4 consecutive k of row r per 32-wide k block.
// Only k blocks up to the end of the row's 256-row block are written (the sweep never reads beyond).  The padding of
// the factor workspace (rows / columns >= N carry the identity) is masked to zero, as in the transposed f64 operand.
template <int NS>
__global__ __launch_bounds__(256) void w_digits_kernel(const double* __restrict__ W, int64_t N, int64_t Npad,
                                                       double* __restrict__ rs, unsigned char* __restrict__ Wq) {
  const int tid = threadIdx.x, r = tid >> 3, c = tid & 7;
  const int64_t i0 = (int64_t)blockIdx.x * 32, i = i0 + r;
  const int64_t kb_end = ((i0 + 31) / 256 + 1) * 8;  // k blocks of 32 inside the row's 256-row block and before
  const double* row = W + i * Npad;
  double amax = 0.0;
  for (int64_t kb = 0; kb * 32 <= i0 + 31; ++kb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t k = kb * 32 + 4 * c + j;
      if (i < N && k < N) amax = fmax(amax, fabs(row[k]));
    }
  }
  amax = fmax(amax, __shfl_xor(amax, 1, 64));
  amax = fmax(amax, __shfl_xor(amax, 2, 64));
  amax = fmax(amax, __shfl_xor(amax, 4, 64));
  // tight scale: the balanced digits of rint(x / S 2^(8 NS - 1)) reach 0.99609 2^(8 NS - 1) (0x7f7f..7f), so
  // S = (1 + 2^-7) max |W_ik| is enough -- half the scale of rounds 2 / 3, and the dominant error of the scheme (the
  // dropped digit pairs s + s' = NS) is proportional to S_i S'
  const double S = amax > 0.0 ? I8_TIGHT * amax : 1.0;
  __shared__ double wrow[32];
  if (c == 0) {
    rs[i] = S;
    wrow[r] = i < N ? S * S * (double)(i + 1) : 0.0;  // weight of row i in the error model (SweepArgs::rep_ub)
  }
  __syncthreads();
  if (tid == 0) {  // the sweep prices a 32-row fragment at its largest row weight (one multiply-add per fragment)
    double wmax = 0.0;
    for (int q = 0; q < 32; ++q) wmax = fmax(wmax, wrow[q]);
    rs[Npad + blockIdx.x] = wmax;
  }
  const double to_fixed = (NS == 4 ? 2147483648.0 : 549755813888.0) / S;  // 2^(8 NS - 1) / S
  const size_t plane = (size_t)Npad * (size_t)Npad;
  for (int64_t kb = 0; kb < kb_end; ++kb) {
    uint32_t p[NS];
#pragma unroll
    for (int sdx = 0; sdx < NS; ++sdx) p[sdx] = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t k = kb * 32 + 4 * c + j;
      long long q = (i < N && k < N) ? (long long)rint(row[k] * to_fixed) : 0;  // |.| <= 2^(8 NS - 1) / I8_TIGHT
#pragma unroll
      for (int sdx = NS - 1; sdx > 0; --sdx) {
        const int dg = (int)((q + 128) & 255) - 128;
        q = (q - dg) >> 8;
        p[sdx] |= (uint32_t)(dg & 255) << (8 * j);
      }
      p[0] |= (uint32_t)((int)q & 255) << (8 * j);
    }
    unsigned char* dst = Wq + ((size_t)kb * Npad + i) * 32 + 4 * c;
#pragma unroll
    for (int sdx = 0; sdx < NS; ++sdx) *(uint32_t*)(dst + sdx * plane) = p[sdx];
  }
}
// max |W_ik| over the N x N lower triangle -> *out (a non-negative double: its bit pattern orders like an unsigned
// integer, so one 64-bit atomicMax per wave does the reduction).  *out must be zero before the launch.
__global__ __launch_bounds__(256) void w_absmax_kernel(const double* __restrict__ W, int64_t N, int64_t Npad,
                                                       unsigned long long* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double amax = 0.0;
  if (i < N) {
    const double* row = W + i * Npad;
    for (int64_t k = lane; k <= i; k += 64) amax = fmax(amax, fabs(row[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o, 64));
  if (lane == 0 && amax > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(amax));
}
void launch_w_absmax(hipStream_t s, const double* W, int64_t N, int64_t Npad, double* out) {
  hipLaunchKernelGGL(w_absmax_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, N, Npad,
                     (unsigned long long*)out);
}

void launch_w_digits(hipStream_t s, const double* W, int64_t N, int64_t Npad, double* rs, void* Wq, int planes) {
  if (planes == 5)
    hipLaunchKernelGGL(w_digits_kernel<5>, dim3((unsigned)(Npad / 32)), dim3(256), 0, s, W, N, Npad, rs, (unsigned char*)Wq);
  else
    hipLaunchKernelGGL(w_digits_kernel<4>, dim3((unsigned)(Npad / 32)), dim3(256), 0, s, W, N, Npad, rs, (unsigned char*)Wq);
}

// The training rows of the int8 sweep's generating steps as DMA-able tiles (tgp_kernels_sweep_i8.inc): tile t =
// [32 rows of Xs][dp], alpha[32], |row|^2 [32], zero padded to xt doubles (whole KiB).
__global__ __launch_bounds__(256) void xs_tiles_kernel(const double* __restrict__ Xs, const double* __restrict__ alpha, int dp, int xt,
                                                       double* __restrict__ out) {
  const int64_t tile = blockIdx.x;
  for (int e = threadIdx.x; e < xt; e += 256) {
    double v = 0.0;
    if (e < 32 * dp) v = Xs[tile * 32 * dp + e];
    else if (e < 32 * dp + 32) v = alpha[tile * 32 + (e - 32 * dp)];
    else if (e < 32 * dp + 64) {   // |row|^2: the dot-product form of the distances (generation on the matrix core)
      const double* row = Xs + (tile * 32 + (e - 32 * dp - 32)) * dp;
      for (int c = 0; c < dp; ++c) v = fma(row[c], row[c], v);
    }
    out[tile * xt + e] = v;
  }
}
void launch_xs_tiles(hipStream_t s, const double* Xs, const double* alpha, int64_t Npad, int dp, int xt, double* out) {
  hipLaunchKernelGGL(xs_tiles_kernel, dim3((unsigned)(Npad / 32)), dim3(256), 0, s, Xs, alpha, dp, xt, out);
}

// ---------------------------------------------------------------------------------------------
// Cross-device merge of per-shard winners (host mirror: trieste_amd/distributed.py merge_best):
// gathered [P][2][V] -- rank p's V values followed by its V global indices (int64 bit patterns in the 8-byte
// slots, exactly what the all-gather moved).  Larger value wins (smaller if `minimize`), ties go to the
// smaller global index (tf.math.argmax / argmin over the unsharded set); a NaN value or an index < 0 / the
// INT64_MAX "nothing found" mark never wins.  No valid entry: (NaN, -1).  One thread per v, fixed order.
__global__ void merge_winners_kernel(const double* __restrict__ gathered, int P, int V, int minimize,
                                     double* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double bv = -INFINITY, braw = __builtin_nan("");
  int64_t bi = INT64_MAX;
  for (int p = 0; p < P; ++p) {
    const double raw = gathered[((size_t)p * 2 + 0) * V + v];
    const int64_t idx = ((const int64_t*)gathered)[((size_t)p * 2 + 1) * V + v];
    if (raw != raw || idx < 0 || idx == INT64_MAX) continue;
    const double key = minimize ? -raw : raw;
    if (bi == INT64_MAX || better(key, idx, bv, bi)) {
      bv = key;
      bi = idx;
      braw = raw;
    }
  }
  out[v] = braw;
  ((int64_t*)out)[V + v] = bi == INT64_MAX ? -1 : bi;
}
// ---------------------------------------------------------------------------------------------
// A-posteriori repair of the split-precision sweep (tgp_api.hip sweep_i8_repaired).  The int8
// kernel leaves ub [M]: +inf where the candidate's own truncation bound exceeds the parity tolerance, else the upper
// end of the interval its acquisition value lies in; *L = the largest LOWER end over the sweep.  Whatever could still
// be the float64 arg-max (ub >= L) and whatever violates the tolerance goes on a list, is recomputed by the float64
// sweep (SPLIT instantiation, count on the device) and scattered back.
// stats (tgp_internal.hpp RS_*): {count, M, epoch tag, uniform-canary violations, checked, worst |d var| / bound (bits of a
// double), adversarial violations, checked, [L slot], [route], adversarial worst ratio, samples only the slack saved}; the
// canary words accumulate over the sweeps of one rung of the ladder (reset_canary: the rung is new)
__global__ void repair_begin_kernel(int64_t* stats, int64_t M, int64_t tag, int reset_canary) {
  stats[RS_COUNT] = 0;
  stats[RS_M] = M;
  stats[RS_TAG] = tag;
  if (reset_canary)
    stats[RS_VIOL] = stats[RS_CHECKED] = stats[RS_WORST] = stats[RS_ADV_VIOL] = stats[RS_ADV_CHECKED] = stats[RS_ADV_WORST] =
        stats[RS_SLACK_SAVED] = 0;
}
void launch_repair_begin(hipStream_t s, int64_t* stats, int64_t M, int64_t tag, bool reset_canary) {
  hipLaunchKernelGGL(repair_begin_kernel, dim3(1), dim3(1), 0, s, stats, M, tag, reset_canary ? 1 : 0);
}
__global__ __launch_bounds__(256) void repair_flag_kernel(const double* __restrict__ ub, int64_t M, const double* L,
                                                          int64_t* __restrict__ list, int64_t* stats, int64_t canary_off) {
  const double Lv = L ? *L : INFINITY;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < M; i0 += stride) {
    const int64_t i = i0 + threadIdx.x;
    bool flag = false;
    if (i < M) {
      const double u = ub[i];
      flag = u == INFINITY || (Lv > -INFINITY && Lv < INFINITY && u >= Lv) ||
             (canary_off >= 0 && ((i + canary_off) & (I8_CANARY_PERIOD - 1)) == 0);
    }
    // wave-aggregated append: one atomic per wave
    const unsigned long long mask = __ballot(flag);
    if (mask) {
      const int n = __popcll(mask);
      unsigned long long base = 0;
      if (lane == __ffsll((long long)mask) - 1) base = atomicAdd((unsigned long long*)stats, (unsigned long long)n);
      base = __shfl((long long)base, __ffsll((long long)mask) - 1, 64);
      if (flag) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    }
  }
}
void launch_repair_flag(hipStream_t s, const double* ub, int64_t M, const double* L, int64_t* list, int64_t* stats,
                        int64_t canary_off) {
  const int64_t blocks = std::min<int64_t>((M + 255) / 256, 1024);
  hipLaunchKernelGGL(repair_flag_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ub, M, L, list, stats, canary_off);
}
// The canary of TGP_PREC_AUTO: the sampled candidates were recomputed in float64 with the repair list; |var_f64 - var_int8|
// against the bound the int8 kernel priced that candidate at.  `slack` covers the float64 kernels' own rounding (their
// summation orders differ; ADVICE r05: it scales with the sweep's rounding floor, and what only the slack saved is counted).
// Two strata, reported apart: the UNIFORM sample (one candidate in 4096, list entries recognised by their index) and the
// ADVERSARIAL one (adv_sel: per 1 / 64 of the sweep the unflagged candidate whose bound sits closest to its tolerance).
__device__ __forceinline__ void canary_compare(int64_t* stats, double v64, double v8, double bound, double slack, int w_viol,
                                               int w_checked, int w_worst) {
  const double dv = fabs(v64 - v8);
  atomicAdd((unsigned long long*)(stats + w_checked), 1ull);
  if (!(dv <= bound + slack)) atomicAdd((unsigned long long*)(stats + w_viol), 1ull);
  else if (!(dv <= bound)) atomicAdd((unsigned long long*)(stats + RS_SLACK_SAVED), 1ull);
  const double ratio = dv / fmax(bound + slack, 1e-300);   // >= 0: the bit patterns order like the values
  atomicMax((unsigned long long*)(stats + w_worst), (unsigned long long)__double_as_longlong(ratio));
}
__global__ __launch_bounds__(256) void repair_canary_kernel(const int64_t* __restrict__ list, int64_t* stats,
                                                            const double* __restrict__ rvar, const double* __restrict__ rec,
                                                            int64_t canary_off, double slack,
                                                            const double* __restrict__ adv_sel) {
  const int64_t n = stats[RS_COUNT];
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
    const int64_t j = list[r];
    if (((j + canary_off) & (I8_CANARY_PERIOD - 1)) != 0) continue;
    const double* c = rec + 2 * ((j + canary_off) / I8_CANARY_PERIOD);
    canary_compare(stats, rvar[r], c[0], c[1], slack, RS_VIOL, RS_CHECKED, RS_WORST);
  }
  if (adv_sel && blockIdx.x == 0 && threadIdx.x < I8_ADV_GROUPS) {
    const double* c = adv_sel + 3 * threadIdx.x;
    const int64_t pos = (int64_t)__double_as_longlong(c[0]);
    if (pos >= 0 && pos < n) canary_compare(stats, rvar[pos], c[1], c[2], slack, RS_ADV_VIOL, RS_ADV_CHECKED, RS_ADV_WORST);
  }
}
void launch_repair_canary(hipStream_t s, const int64_t* list, int64_t* stats, int64_t cap, const double* rvar,
                          const double* rec, int64_t canary_off, double slack, const double* adv_sel) {
  const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((cap + 255) / 256, 256));
  hipLaunchKernelGGL(repair_canary_kernel, dim3((unsigned)blocks), dim3(256), 0, s, list, stats, rvar, rec, canary_off, slack,
                     adv_sel);
}
// the adversarial stratum: one wave per group of consecutive block records; the group's worst (largest bound / tolerance,
// lowest index on ties) unflagged candidate goes on the repair list
__global__ __launch_bounds__(64) void repair_adv_kernel(const double* __restrict__ adv_rec, int64_t nblk, int groups,
                                                        int64_t* __restrict__ list, int64_t* stats, double* __restrict__ adv_sel) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (g >= groups) {   // fewer blocks than groups: the slots beyond carry "nothing"
    if (lane == 0) {
      adv_sel[3 * g] = __longlong_as_double(-1ll);
      adv_sel[3 * g + 1] = adv_sel[3 * g + 2] = 0.0;
    }
    return;
  }
  const int64_t b0 = nblk * g / groups, b1 = nblk * (g + 1) / groups;
  double r = -1.0;
  int64_t at = INT64_MAX;   // the record (block) the best ratio came from
  for (int64_t b = b0 + lane; b < b1; b += 64) {
    const double x = adv_rec[4 * b];
    if (x >= 0.0 && better(x, b, r, at)) {
      r = x;
      at = b;
    }
  }
  wave_argmax(r, at);
  if (lane == 0) {
    double* sel = adv_sel + 3 * g;
    if (r >= 0.0 && at != INT64_MAX) {
      const double* rec = adv_rec + 4 * at;
      const unsigned long long pos = atomicAdd((unsigned long long*)(stats + RS_COUNT), 1ull);
      list[pos] = (int64_t)__double_as_longlong(rec[1]);
      sel[0] = __longlong_as_double((long long)pos);
      sel[1] = rec[2];
      sel[2] = rec[3];
    } else {
      sel[0] = __longlong_as_double(-1ll);
      sel[1] = sel[2] = 0.0;
    }
  }
}
void launch_repair_adv(hipStream_t s, const double* adv_rec, int64_t nblk, int64_t* list, int64_t* stats, double* adv_sel) {
  const int groups = (int)std::min<int64_t>(I8_ADV_GROUPS, nblk);
  hipLaunchKernelGGL(repair_adv_kernel, dim3(I8_ADV_GROUPS), dim3(64), 0, s, adv_rec, nblk, groups, list, stats, adv_sel);
}
__global__ __launch_bounds__(256) void repair_gather_kernel(const double* __restrict__ Xq, int d,
                                                            const int64_t* __restrict__ list, const int64_t* count,
                                                            double* __restrict__ Xg) {
  const int64_t n = *count * d;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / d;
    Xg[e] = Xq[list[r] * d + (e - r * d)];
  }
}
void launch_repair_gather(hipStream_t s, const double* Xq, int d, const int64_t* list, const int64_t* count, int64_t cap,
                          double* Xg) {
  const int64_t blocks = std::min<int64_t>((cap * d + 255) / 256, 1024);
  hipLaunchKernelGGL(repair_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Xq, d, list, count, Xg);
}
__global__ __launch_bounds__(256) void repair_scatter_kernel(const int64_t* __restrict__ list, const int64_t* count,
                                                             const double* rmean, const double* rvar, const double* racq,
                                                             double* mean, double* var, double* acq,
                                                             const double* __restrict__ ub, const double* L) {
  const int64_t n = *count;
  const double Lv = L ? *L : INFINITY;
  const bool band = Lv > -INFINITY && Lv < INFINITY;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
    const int64_t j = list[r];
    const double u = ub[j];
    if (!(u == INFINITY || (band && u >= Lv))) continue;   // a sample of the canary: compared, not scattered
    if (mean) mean[j] = rmean[r];
    if (var) var[j] = rvar[r];
    if (acq) acq[j] = racq[r];
  }
}
void launch_repair_scatter(hipStream_t s, const int64_t* list, const int64_t* count, int64_t cap, const double* rmean,
                           const double* rvar, const double* racq, double* mean, double* var, double* acq, const double* ub,
                           const double* L) {
  const int64_t blocks = std::min<int64_t>((cap + 255) / 256, 1024);
  hipLaunchKernelGGL(repair_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, s, list, count, rmean, rvar, racq, mean,
                     var, acq, ub, L);
}
__global__ __launch_bounds__(256) void values_argmax_kernel(const double* __restrict__ vals, int64_t M, int64_t index_base,
                                                            double* blk_val, int64_t* blk_idx, int64_t nslots) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  double v = -INFINITY;
  int64_t i = INT64_MAX;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < M; t += (int64_t)gridDim.x * 256) {
    const double x = vals[t];
    if (!(x != x) && better(x, index_base + t, v, i)) {
      v = x;
      i = index_base + t;
    }
  }
  wave_argmax(v, i);
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (better(sv[w], si[w], v, i)) {
        v = sv[w];
        i = si[w];
      }
    blk_val[blockIdx.x] = v;
    blk_idx[blockIdx.x] = i;
  }
  // the slots beyond the grid carry "nothing found"
  for (int64_t t = (int64_t)gridDim.x + (int64_t)blockIdx.x * 256 + threadIdx.x; t < nslots; t += (int64_t)gridDim.x * 256) {
    blk_val[t] = -INFINITY;
    blk_idx[t] = INT64_MAX;
  }
}
void launch_values_argmax(hipStream_t s, const double* vals, int64_t M, int64_t index_base, double* blk_val,
                          int64_t* blk_idx, int64_t nslots) {
  const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((M + 255) / 256, 256), nslots));
  hipLaunchKernelGGL(values_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, vals, M, index_base, blk_val, blk_idx,
                     nslots);
}

void launch_merge_winners(hipStream_t s, const double* gathered, int P, int V, int minimize, double* out) {
  hipLaunchKernelGGL(merge_winners_kernel, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, s, gathered, P, V,
                     minimize, out);
}

// ---------------------------------------------------------------------------------------------
// TGP_PREC_AUTO, the repair of a FEW candidates as a product (round 5).  The SPLIT sweep recomputes a handful of candidates
// in the latency of its heaviest row group's walk (N = 4096: the last row block alone is 12 % of the triangle on ONE CU,
// 0.9 - 1.2 ms).  Up to `pcap` candidates go through building blocks that spread over the chip instead: B = K*^T
// [Npad][Ppad] (kstar_t with the count on the device), C = W B (gemm_tall of tgp_api.hip), then column sums
// (k*.alpha, sum c^2) in two deterministic steps and the common tail.  Reference: the same posterior as the sweep,
// models/gpflow/interface.py:119-124.  Nothing here reads the count on the host: repair_route_kernel hands it to exactly
// one of the two paths (the other one sees zero candidates).
__global__ void repair_route_kernel(const int64_t* __restrict__ stats, int64_t pcap, int64_t* __restrict__ route) {
  const int64_t c = stats[0];
  route[0] = c > pcap ? c : 0;    // the SPLIT sweep's count
  route[1] = c <= pcap ? c : 0;   // the product path's count
}
void launch_repair_route(hipStream_t s, const int64_t* stats, int64_t pcap, int64_t* route) {
  hipLaunchKernelGGL(repair_route_kernel, dim3(1), dim3(1), 0, s, stats, pcap, route);
}

// B[k][p] = k(X_k, x_p), k-major [Npad][Ppad]; columns from *P_dev on are zero (kstar_t_kernel of tgp_kernels_grad.hip
// with the count on the device)
__global__ void kstar_t_dev_kernel(ModelDev m, const double* __restrict__ Xq, const int64_t* __restrict__ P_dev,
                                   int64_t Ppad, double* __restrict__ B) {
  const int64_t p = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t k = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (p >= Ppad || k >= m.Npad) return;
  const int64_t P = *P_dev;
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
void launch_kstar_t_dev(hipStream_t s, const ModelDev& m, const double* Xq, const int64_t* P_dev, int64_t Ppad, double* B) {
  dim3 grid((unsigned)(Ppad / 64), (unsigned)(m.Npad / 4));
  hipLaunchKernelGGL(kstar_t_dev_kernel, grid, dim3(256), 0, s, m, Xq, P_dev, Ppad, B);
}

// part[chunk][0][p] = sum_{k in chunk} B[k][p] alpha[k],  part[chunk][1][p] = sum_{k in chunk} C[k][p]^2:
// block (column group of 64, row chunk), thread (column, row lane of 4); the four lanes' sums are added in lane order
constexpr int REPAIR_CHUNKS = 32;
__global__ __launch_bounds__(256) void repair_colsums_kernel(const double* __restrict__ B, const double* __restrict__ C,
                                                             const double* __restrict__ alpha, int64_t Npad, int64_t Ppad,
                                                             double* __restrict__ part) {
  __shared__ double sm[4][64], sq[4][64];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * 64 + col;
  const int64_t rows = Npad / REPAIR_CHUNKS, r0 = (int64_t)blockIdx.y * rows;
  double m = 0.0, q = 0.0;
  for (int64_t r = r0 + rl; r < r0 + rows; r += 4) {
    const double c = C[r * Ppad + p];
    m = fma(B[r * Ppad + p], alpha[r], m);
    q = fma(c, c, q);
  }
  sm[rl][col] = m;
  sq[rl][col] = q;
  __syncthreads();
  if (rl == 0) {
    part[((size_t)blockIdx.y * 2 + 0) * Ppad + p] = ((sm[0][col] + sm[1][col]) + sm[2][col]) + sm[3][col];
    part[((size_t)blockIdx.y * 2 + 1) * Ppad + p] = ((sq[0][col] + sq[1][col]) + sq[2][col]) + sq[3][col];
  }
}
// the chunks in order, then the sweep's own tail (sweep_combine_kernel above): clip, acquisition value, outputs
__global__ __launch_bounds__(256) void repair_tail_kernel(ModelDev md, const double* __restrict__ part, int64_t Ppad,
                                                          const int64_t* __restrict__ P_dev, int acq_kind, double acq_param,
                                                          double* __restrict__ mean_out, double* __restrict__ var_out,
                                                          double* __restrict__ acq_out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= *P_dev || p >= Ppad) return;
  double m = 0.0, q = 0.0;
  for (int c = 0; c < REPAIR_CHUNKS; ++c) {
    m += part[((size_t)c * 2 + 0) * Ppad + p];
    q += part[((size_t)c * 2 + 1) * Ppad + p];
  }
  const double mean = m + md.mean_const;
  const double var = fmax(md.variance - q, VAR_FLOOR);
  if (mean_out) mean_out[p] = mean;
  if (var_out) var_out[p] = var;
  if (acq_kind >= 0 && acq_out) acq_out[p] = acq_tail(acq_kind, acq_param, mean, var, md.noise);
}
void launch_repair_product_tail(hipStream_t s, const ModelDev& m, const double* B, const double* C, int64_t Ppad,
                                const int64_t* P_dev, int acq_kind, double acq_param, double* part, double* mean_out,
                                double* var_out, double* acq_out) {
  hipLaunchKernelGGL(repair_colsums_kernel, dim3((unsigned)(Ppad / 64), REPAIR_CHUNKS), dim3(256), 0, s, B, C, m.alpha,
                     m.Npad, Ppad, part);
  hipLaunchKernelGGL(repair_tail_kernel, dim3((unsigned)((Ppad + 255) / 256)), dim3(256), 0, s, m, part, Ppad, P_dev,
                     acq_kind, acq_param, mean_out, var_out, acq_out);
}
int64_t repair_product_part_doubles(int64_t Ppad) { return (int64_t)REPAIR_CHUNKS * 2 * Ppad; }

}  // namespace tgp
