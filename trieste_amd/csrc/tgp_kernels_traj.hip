// Decoupled Thompson trajectories on gfx950.
//   f_b(x) = sum_f phi_f(x) ws[f][b] + sum_k k(x, X_k) v[k][b] + c,   phi_f(x) = cos((x / ls) . W_f + b_f)
// (the sqrt(2 variance / F) factor is folded into ws).  == ResampleableDecoupledFeatureFunctions.call
// (reference sampler.py:846-855) + gpflux RandomFourierFeaturesCosine + feature_decomposition_
// trajectory.__call__ (sampler.py:923-936).  One thread per candidate; basis rows, training rows and
// weights come through scalar loads; the [M, F + N] feature matrix is never materialised (SURVEY
// K10, K11).  The kernel is fp64-VALU bound, so the per-pair instruction count is what matters:
// distances use r2 = |a|^2 + |b|^2 - 2 a.b (D fused multiply-adds instead of 2 D), sqrt / exp / cos
// are the branch-free forms of tgp_dev.hpp.
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

namespace tgp {

constexpr int TRAJ_MAXB = 16;

// (the kernel evaluation of the trajectory loops -- TrajShape, traj_sqrt, traj_exp2, traj_shape -- lives in tgp_dev.hpp since
// round 6: the int8 sweep's generating steps use the same forms)
// cos(x . W + b) from y = (x . W + b) / pi + 1/2 (the basis is stored in half turns): cos(theta) = sin(pi y) =
// (-1)^n sin(pi f), n = rint(y), f = y - n exact, |f| <= 1/2.  sin(pi f) = f P(f^2) with a degree-8 minimax-fitted P
// (truncation 3.9e-17 relative); the sign goes onto f (odd function) as one xor of its high word.  15 instructions
// against ~30 for the Cody-Waite reduction by pi / 2 + two polynomials + quadrant selects of fast_cos.
__device__ __forceinline__ double traj_cos_halfturns(double y) {
  const double n = rint(y);
  double f = y - n;
  const unsigned flip = (unsigned)(int)n << 31;
  f = __hiloint2double(__double2hiint(f) ^ (int)flip, __double2loint(f));
  const double z = f * f;
  double p = 7.697847275590284e-07;
  p = fma(p, z, -2.1903499125519212e-05);
  p = fma(p, z, 0.00046629981702308817);
  p = fma(p, z, -0.007370430506093225);
  p = fma(p, z, 0.08214588657312355);
  p = fma(p, z, -0.599264529318946);
  p = fma(p, z, 2.5501640398773007);
  p = fma(p, z, -5.167712780049969);
  p = fma(p, z, 3.141592653589793);
  return f * p;
}
// PER_TRAJ: logical item = (candidate j, trajectory b) with its own input row (BP = 1); else item = j and the BP
// accumulators are the trajectories.  EXACT: B == BP -- the accumulator updates carry no test of b against B.  Both are
// template parameters because as run-time tests they sat INSIDE the two inner loops as wave-uniform branches, each
// update with a scalar load and a wait of its own (round 2's form: ~60 issued instructions per kernel evaluation for
// 45 of arithmetic).
template <int KIND, int DP, int BP, bool EXACT, bool PER_TRAJ>
__global__ __launch_bounds__(256) void traj_eval_kernel(TrajDev t, const double* __restrict__ Xq,
                                                        int64_t M, int rff_only,
                                                        double* __restrict__ out,
                                                        double* __restrict__ blk_val,
                                                        int64_t* __restrict__ blk_idx,
                                                        int64_t index_base) {
  static_assert(!PER_TRAJ || BP == 1, "one accumulator per (candidate, trajectory) item");
  const int B = t.B, d = t.m.d;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nitems = PER_TRAJ ? M * B : M;
  const bool valid = item < nitems;
  const int myb = PER_TRAJ ? (int)(item % B) : 0;
  double xq[DP];
  double nb = 0.0;
#pragma unroll
  for (int c = 0; c < DP; ++c) {
    xq[c] = (c < d && valid) ? Xq[item * d + c] / as_const(t.m.ls)[c] : 0.0;
    nb = fma(xq[c], xq[c], nb);
  }
  double acc[BP];
#pragma unroll
  for (int b = 0; b < BP; ++b) acc[b] = 0.0;

  const cptr W = as_const(t.rffW_ht);   // the basis in half turns (W / pi, b / pi + 1/2)
  const cptr bb = as_const(t.rffb_ht);
  const cptr ws = as_const(t.ws);
#pragma unroll 1  // (two features per iteration need > 100 SGPRs: the spill reloads cost more than the interleaving gains)
  for (int f = 0; f < t.F; ++f) {
    double arg = bb[f];
#pragma unroll
    for (int c = 0; c < DP; ++c) arg = fma(xq[c], W[(int64_t)f * DP + c], arg);
    const double ph = traj_cos_halfturns(arg);
    if (PER_TRAJ) {
      acc[0] = fma(ph, t.ws[(int64_t)f * B + myb], acc[0]);
    } else {
#pragma unroll
      for (int b = 0; b < BP; ++b)
        if (EXACT || b < B) acc[b] = fma(ph, ws[(int64_t)f * (EXACT ? BP : B) + b], acc[b]);
    }
  }
  if (rff_only == 0 || rff_only == 3) {
    const cptr xs = as_const(t.m.Xs);
    const cptr xn = as_const(t.m.xn);
    const cptr vv = as_const(t.v);
    constexpr double SC = TrajShape<KIND>::SCALE;
    const double nbs = SC * nb;
    double acck[BP];  // sum_k shape_k v[k][b]: the variance multiplies it once, below
#pragma unroll
    for (int b = 0; b < BP; ++b) acck[b] = 0.0;
#pragma unroll 2
    for (int64_t k = 0; k < t.m.N; ++k) {
      double q;
      if constexpr (KIND == KIND_M12) {
        // exp(-r) has a kink at r = 0: the dot-product form's cancellation error (~1e-15 on r^2, i.e. 3e-8 on r) would
        // show at candidates that coincide with training inputs -- the difference form is exact there (2 D instructions)
        double r2 = 0.0;
#pragma unroll
        for (int c = 0; c < DP; ++c) {
          const double t0 = xq[c] - xs[k * DP + c];
          r2 = fma(t0, t0, r2);
        }
        q = SC * r2;
      } else {
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < DP; ++c) dot = fma(xq[c], xs[k * DP + c], dot);
        q = fma(-2.0 * SC, dot, fma(SC, xn[k], nbs));   // SCALE (|x|^2 + |X_k|^2 - 2 x . X_k)
      }
      const double kv = traj_shape<KIND>(q);
      if (PER_TRAJ) {
        acck[0] = fma(kv, t.v[k * B + myb], acck[0]);
      } else {
#pragma unroll
        for (int b = 0; b < BP; ++b)
          if (EXACT || b < B) acck[b] = fma(kv, vv[k * (EXACT ? BP : B) + b], acck[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < BP; ++b) acc[b] = fma(t.m.variance, acck[b], acc[b]);
  }
  // 1: bare projection Phi w; 2: RFF trajectory (+ mean); 3: bare kernel sums sum_k k(x, X_k) v[k][b]
  const double c0 = (rff_only == 1 || rff_only == 3) ? 0.0 : t.m.mean_const;
  if (out && valid) {
    if (PER_TRAJ) out[item] = acc[0] + c0;
    else {
#pragma unroll
      for (int b = 0; b < BP; ++b)
        if (EXACT || b < B) out[item * B + b] = acc[b] + c0;
    }
  }
  if (!PER_TRAJ && blk_val) {  // per-workgroup arg-min per trajectory (shared-input mode only)
    __shared__ double wv[4][TRAJ_MAXB];
    __shared__ int64_t wi[4][TRAJ_MAXB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < BP; ++b) {
      if (EXACT || b < B) {
        double v = valid ? -(acc[b] + c0) : -INFINITY;  // arg-min == arg-max of the negation
        if (v != v) v = -INFINITY;
        int64_t i = valid ? index_base + item : INT64_MAX;
        wave_argmax(v, i);
        if (lane == 0) {
          wv[w][b] = v;
          wi[w][b] = i;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < B) {
      const int b = threadIdx.x;
      double v = wv[0][b];
      int64_t i = wi[0][b];
      for (int ww = 1; ww < 4; ++ww)
        if (better(wv[ww][b], wi[ww][b], v, i)) {
          v = wv[ww][b];
          i = wi[ww][b];
        }
      blk_val[(int64_t)blockIdx.x * B + b] = -v;
      blk_idx[(int64_t)blockIdx.x * B + b] = i;
    }
  }
}

template <int KIND, int DP>
static void launch_traj_bp(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, int per_traj,
                           int rff_only, double* out, double* bv, int64_t* bi, int64_t base) {
  const int64_t nitems = per_traj ? M * t.B : M;
  dim3 g((unsigned)((nitems + 255) / 256)), b(256);
#define TGP_TRAJ_LAUNCH(BP, EXACT, PT) \
  hipLaunchKernelGGL((traj_eval_kernel<KIND, DP, BP, EXACT, PT>), g, b, 0, s, t, Xq, M, rff_only, out, bv, bi, base)
  if (per_traj) TGP_TRAJ_LAUNCH(1, true, true);
  else if (t.B == 1) TGP_TRAJ_LAUNCH(1, true, false);
  else if (t.B == 4) TGP_TRAJ_LAUNCH(4, true, false);
  else if (t.B < 4) TGP_TRAJ_LAUNCH(4, false, false);
  else if (t.B == 16) TGP_TRAJ_LAUNCH(16, true, false);
  else TGP_TRAJ_LAUNCH(16, false, false);
#undef TGP_TRAJ_LAUNCH
}

template <int KIND>
static void launch_traj_dp(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, int per_traj,
                           int rff_only, double* out, double* bv, int64_t* bi, int64_t base) {
  switch (t.m.dp) {
    case 2: launch_traj_bp<KIND, 2>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    case 4: launch_traj_bp<KIND, 4>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    case 6: launch_traj_bp<KIND, 6>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    case 8: launch_traj_bp<KIND, 8>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    case 16: launch_traj_bp<KIND, 16>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    default: launch_traj_bp<KIND, 32>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
  }
}

static void launch_traj_any(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, int per_traj,
                            int rff_only, double* out, double* bv, int64_t* bi, int64_t base) {
  switch (t.m.kind) {
    case KIND_RBF: launch_traj_dp<KIND_RBF>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    case KIND_M12: launch_traj_dp<KIND_M12>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    case KIND_M32: launch_traj_dp<KIND_M32>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
    default: launch_traj_dp<KIND_M52>(s, t, Xq, M, per_traj, rff_only, out, bv, bi, base); break;
  }
}

int64_t traj_grid(int64_t M) { return (M + 255) / 256; }

// out[M][B] = sum_k k(x, X_k) v[k][b] for B <= 16 weight vectors over the model's training inputs (t.F must be 0):
// the m = rank-of-the-update dot products GIBBON's conditioned variance needs per candidate.
void launch_kernel_sums(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, double* out) {
  launch_traj_any(s, t, Xq, M, 0, 3, out, nullptr, nullptr, 0);
}

void launch_traj_eval(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, int per_traj,
                      double* out, double* blk_val, int64_t* blk_idx, int64_t index_base) {
  launch_traj_any(s, t, Xq, M, per_traj, t.canonical ? 0 : 2, out, blk_val, blk_idx, index_base);
}

// Value and gradient of trajectory b at its own point x[p][b][:] -- the pair the L-BFGS-B
// refinement of Greedy/ParallelContinuousThompsonSampling needs (reference
// continuous_thompson_sampling.py:30-245 via acquisition/optimizer.py:628-629, where TF autodiff
// produces it).  One workgroup per (point, trajectory); threads stride over the F features and
// the N training rows; d phi_f / dx = -sin(arg) W_f / ls,  d k(x, X_k) / dx = 2 k'(r2) (x - X_k) / ls^2
// in scaled coordinates.  A few hundred items per L-BFGS-B iteration: latency, not throughput.
template <int DP>
__global__ __launch_bounds__(256) void traj_grad_kernel(TrajDev t, const double* __restrict__ Xq,
                                                        int64_t nitems, double* __restrict__ val,
                                                        double* __restrict__ grad) {
  __shared__ double red[4][DP + 1];
  const int64_t item = blockIdx.x;
  const int B = t.B, d = t.m.d, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = (int)(item % B);
  double xq[DP], g[DP];
#pragma unroll
  for (int c = 0; c < DP; ++c) {
    xq[c] = (c < d) ? Xq[item * d + c] / t.m.ls[c] : 0.0;
    g[c] = 0.0;
  }
  double acc = 0.0;
  for (int f = tid; f < t.F; f += 256) {
    double arg = t.rffb[f];
#pragma unroll
    for (int c = 0; c < DP; ++c) arg = fma(xq[c], t.rffW[(int64_t)f * DP + c], arg);
    const double wv = t.ws[(int64_t)f * B + b];
    acc = fma(cos(arg), wv, acc);
    const double sw = -sin(arg) * wv;
#pragma unroll
    for (int c = 0; c < DP; ++c) g[c] = fma(sw, t.rffW[(int64_t)f * DP + c], g[c]);
  }
  for (int64_t k = tid; k < (t.canonical ? t.m.N : 0); k += 256) {
    double r2 = 0.0, df[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) {
      df[c] = xq[c] - t.m.Xs[k * DP + c];
      r2 = fma(df[c], df[c], r2);
    }
    const double vk = t.v[k * B + b];
    acc = fma(kernel_rt(t.m.kind, r2, t.m.variance), vk, acc);
    const double f1 = 2.0 * kernel_dr2(t.m.kind, r2, t.m.variance) * vk;
#pragma unroll
    for (int c = 0; c < DP; ++c) g[c] = fma(f1, df[c], g[c]);
  }
  acc = wave_sum(acc);
#pragma unroll
  for (int c = 0; c < DP; ++c) g[c] = wave_sum(g[c]);
  if (lane == 0) {
    red[w][0] = acc;
#pragma unroll
    for (int c = 0; c < DP; ++c) red[w][1 + c] = g[c];
  }
  __syncthreads();
  if (tid <= d) {
    const double s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid == 0) val[item] = s + t.m.mean_const;
    else grad[item * d + (tid - 1)] = s / t.m.ls[tid - 1];  // back to raw coordinates
  }
}

void launch_traj_grad(hipStream_t s, const TrajDev& t, const double* Xq, int64_t nitems, double* val,
                      double* grad) {
  dim3 g((unsigned)nitems), b(256);
  switch (t.m.dp) {
    case 2: hipLaunchKernelGGL(traj_grad_kernel<2>, g, b, 0, s, t, Xq, nitems, val, grad); break;
    case 4: hipLaunchKernelGGL(traj_grad_kernel<4>, g, b, 0, s, t, Xq, nitems, val, grad); break;
    case 6: hipLaunchKernelGGL(traj_grad_kernel<6>, g, b, 0, s, t, Xq, nitems, val, grad); break;
    case 8: hipLaunchKernelGGL(traj_grad_kernel<8>, g, b, 0, s, t, Xq, nitems, val, grad); break;
    case 16: hipLaunchKernelGGL(traj_grad_kernel<16>, g, b, 0, s, t, Xq, nitems, val, grad); break;
    default: hipLaunchKernelGGL(traj_grad_kernel<32>, g, b, 0, s, t, Xq, nitems, val, grad); break;
  }
}

// Phi[i][f] = sqrt(2 variance / F) cos(Xs_i . W_f + b_f) for the N training rows, [Npad][Fp] zero padded
// (the scaled feature matrix of RandomFourierFeatureTrajectorySampler, sampler.py:541-546 / 570-573).
__global__ void rff_features_kernel(TrajDev t, double scale, int64_t Fp, double* __restrict__ Phi) {
  const int64_t f = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t i = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (f >= Fp || i >= t.m.Npad) return;
  double v = 0.0;
  if (f < t.F && i < t.m.N) {
    double arg = t.rffb[f];
    for (int c = 0; c < t.m.dp; ++c) arg = fma(t.m.Xs[i * t.m.dp + c], t.rffW[f * t.m.dp + c], arg);
    v = scale * cos(arg);
  }
  Phi[i * Fp + f] = v;
}

void launch_rff_features(hipStream_t s, const TrajDev& t, double scale, int64_t Fp, double* Phi) {
  dim3 grid((unsigned)(Fp / 64), (unsigned)(t.m.Npad / 4));
  hipLaunchKernelGGL(rff_features_kernel, grid, dim3(256), 0, s, t, scale, Fp, Phi);
}

// Phi_Z w at a set of RAW points (the training inputs): out [npts][B] = sum_f phi_f(x) ws[f][b]
// (sampler.py:726 `phi_Z @ prior_weights`).
void launch_rff_project(hipStream_t s, const TrajDev& t, const double* X_raw, int64_t npts, double* out) {
  launch_traj_any(s, t, X_raw, npts, 0, 1, out, nullptr, nullptr, 0);
}

}  // namespace tgp
