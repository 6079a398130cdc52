// Device-side helpers shared by the gfx950 kernels of libtgp (stationary kernels, normal cdf/pdf,
// acquisition tails, Philox, wave reductions).  CDNA4 only: 64-lane wavefronts, f64 MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int KIND_RBF = 0, KIND_M12 = 1, KIND_M32 = 2, KIND_M52 = 3;
constexpr int ACQ_EI = 0, ACQ_PI = 1, ACQ_NLCB = 2, ACQ_AEI = 3, ACQ_MES = 4, ACQ_GIBBON = 5, ACQ_LOGYVAR = 6;
constexpr double VAR_FLOOR = 1e-12;  // reference interface.py:123

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16).
// lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// D reg r of lane l is D[row = (l>>4) + 4r][col = l&15]   (f64 layout, not the f32 one).
__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- fast fp64 math for the per-entry kernel evaluation ---------------------------------------
// The libm-grade sqrt/exp hipcc inlines cost ~45 VALU instructions per K* entry, most of it
// special-case handling (scaling, class tests, overflow selects) that cannot trigger here:
// sqrt is only called on [1e-36, 1e300] and exp only on (-inf, 0].  These versions keep full
// double accuracy (<= 2 ulp, checked against the oracle in tests/test_gpu_parity.py) at roughly
// half the instructions.
typedef const __attribute__((address_space(4))) double* cptr;  // read-only data, scalar-loadable
__device__ __forceinline__ cptr as_const(const double* p) {
  return (cptr)(const __attribute__((address_space(1))) double*)(p);
}

__device__ __forceinline__ double fast_sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);  // v_rsq_f64
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double dd = fma(-g, g, x);
  g = fma(dd, h, g);
  dd = fma(-g, g, x);
  g = fma(dd, h, g);
  return g;
}

// sqrt(x) and 1/sqrt(x) together (x > 0, normal range): the coupled Goldschmidt pair g -> sqrt(x),
// h -> 1 / (2 sqrt(x)) with ONE coupled iteration (the pivot chain of the 64-leaf is a serial dependency of its
// wave).  v_rsq_f64 delivers ~2^-26; one Goldschmidt step squares that, the residual step on g brings sqrt(x) to
// ~1 ulp, 1/sqrt(x) to ~2 ulp.
__device__ __forceinline__ void sqrt_and_rsqrt_short(double x, double& g_out, double& rs_out) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double dd = fma(-g, g, x);
  g_out = fma(dd, h, g);
  rs_out = h + h;
}

// exp(x) for x <= 0.  n = rint(x / ln2), r = x - n ln2 (two-term Cody-Waite), degree-11
// Chebyshev-fitted polynomial on |r| <= ln2/2 (relative error 4.2e-18), ldexp.
__device__ __forceinline__ double fast_exp_nonpos(double x) {
  x = fmax(x, -745.5);  // exp(-745.5) underflows to 0 through ldexp
  const double n = rint(x * 1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 2.5110037605963777e-08;
  p = fma(p, r, 2.763263963904103e-07);
  p = fma(p, r, 2.755724091857897e-06);
  p = fma(p, r, 2.4801485482328494e-05);
  p = fma(p, r, 0.00019841269890047113);
  p = fma(p, r, 0.0013888888952314775);
  p = fma(p, r, 0.008333333333319601);
  p = fma(p, r, 0.0416666666664881);
  p = fma(p, r, 0.1666666666666668);
  p = fma(p, r, 0.5000000000000019);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// cos(x) for |x| < 1e6: three-term Cody-Waite reduction by pi/2, degree-13 / degree-12 polynomials on
// |r| <= pi/4 (abs error 1e-17), quadrant select.  About 25 instructions (libm: ~50 with Payne-Hanek
// and special cases that cannot trigger for RFF arguments).
__device__ __forceinline__ double fast_cos(double x) {
  const double n = rint(x * 0.6366197723675814);
  double r = fma(n, -1.5707963109016418, x);
  r = fma(n, -1.5893254712295857e-08, r);
  r = fma(n, -6.123233995736766e-17, r);
  const int q = (int)n;
  const double z = r * r;
  double ps = 1.5918129294866608e-10;
  ps = fma(ps, z, -2.5051131845003624e-08);
  ps = fma(ps, z, 2.755731610255244e-06);
  ps = fma(ps, z, -0.00019841269836758574);
  ps = fma(ps, z, 0.008333333333330948);
  ps = fma(ps, z, -0.16666666666666666);
  const double sv = fma(r * z, ps, r);
  double pc = -1.1382632425521717e-11;
  pc = fma(pc, z, 2.08761462684032e-09);
  pc = fma(pc, z, -2.7557317271729793e-07);
  pc = fma(pc, z, 2.480158729876569e-05);
  pc = fma(pc, z, -0.0013888888888887398);
  pc = fma(pc, z, 0.041666666666666664);
  const double cv = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double v = (q & 1) ? sv : cv;
  return ((q + 1) & 2) ? -v : v;
}

#ifdef TGP_LIBM_MATH
#define TGP_SQRT(x) sqrt(x)
#define TGP_EXPNEG(x) exp(x)
#else
#define TGP_SQRT(x) fast_sqrt_pos(x)
#define TGP_EXPNEG(x) fast_exp_nonpos(x)
#endif

// ---- stationary kernels (gpflow Stationary / IsotropicStationary) -----
// r2 = scaled squared distance (>= 0 by construction: difference form).
template <int KIND>
__device__ __forceinline__ double kernel_from_r2(double r2, double variance) {
  if constexpr (KIND == KIND_RBF) {
    return variance * TGP_EXPNEG(-0.5 * r2);
  } else {
    const double r2c = fmax(r2, 1e-36);  // gpflow: r = sqrt(max(r2, 1e-36))
    const double r = TGP_SQRT(r2c);
    if constexpr (KIND == KIND_M12) {
      return variance * TGP_EXPNEG(-r);
    } else if constexpr (KIND == KIND_M32) {
      const double s = 1.7320508075688772 * r;
      return variance * (1.0 + s) * TGP_EXPNEG(-s);
    } else {
      const double s = 2.23606797749979 * r;
      return variance * fma(5.0 / 3.0, r2c, 1.0 + s) * TGP_EXPNEG(-s);
    }
  }
}

// ---- the SHORT kernel evaluation: the trajectory loops (tgp_kernels_traj.hip, instruction count is the bound: DESIGN.md
// section 4.5) and, since round 6, the generating steps of the int8 sweep (tgp_kernels_sweep_i8.inc) -------------------------
// shape(q) = k / variance as a function of q = SCALE r^2, with SCALE folded into the distance's constants (Matern:
// sqrt(SCALE) r is the argument of both the polynomial and the exponential, so the multiplication by sqrt(3) / sqrt(5)
// disappears); the variance multiplies the finished sum once per candidate instead of every entry; sqrt with ONE
// residual step (~1 ulp) and exp without the underflow clamp (v_ldexp_f64 flushes by itself; the argument is bounded by
// the inputs).  Same accuracy class as kernel_from_r2 (1 - 2 ulp), nine instructions fewer per entry for Matern-5/2.
template <int KIND>
struct TrajShape {
  // q = SCALE r^2 with log2(e) folded in as well (round 5): u = sqrt(q) IS the base-2 exponent of the exponential,
  // exp(-sqrt(c) r) = 2^-u with SCALE = c log2(e)^2 (RBF: exp(-r^2 / 2) = 2^-q with SCALE = log2(e) / 2) -- the
  // multiplication by log2(e) inside the exponential is gone; the polynomial factor takes u / log2(e) through its constants
  static constexpr double L2E = 1.4426950408889634;
  static constexpr double C = KIND == KIND_M32 ? 3.0 : (KIND == KIND_M52 ? 5.0 : 1.0);
  static constexpr double SCALE = KIND == KIND_RBF ? 0.5 * L2E : C * L2E * L2E;
  static constexpr double FLOOR = 1e-36 * SCALE;  // gpflow: r = sqrt(max(r^2, 1e-36))
};
// sqrt(x), x > 0: v_rsq_f64 (2^-26) + ONE coupled Goldschmidt step = 2^-51 relative.  (Rounds 3 / 4 added a residual step
// for the last ulp: two instructions of ~50 per kernel evaluation that a sum of 8192 terms compared at 1e-5 cannot see.)
__device__ __forceinline__ double traj_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double g = x * y, h = 0.5 * y;
  return fma(g, fma(-h, g, 0.5), g);
}
// 2^t for t <= 0: n = rint(t), f = t - n exact, 2^f by a degree-11 fit on |f| <= 1/2 (1.9e-17).
__device__ __forceinline__ double traj_exp2(double t) {
  const double n = rint(t);
  const double f = t - n;
  double p = 4.456675463639861e-10;
  p = fma(p, f, 7.074194562613105e-09);
  p = fma(p, f, 1.0178051192117847e-07);
  p = fma(p, f, 1.3215432534254118e-06);
  p = fma(p, f, 1.5252733856295574e-05);
  p = fma(p, f, 0.00015403530463727982);
  p = fma(p, f, 0.001333355814639035);
  p = fma(p, f, 0.009618129107587253);
  p = fma(p, f, 0.0555041086648217);
  p = fma(p, f, 0.24022650695910158);
  p = fma(p, f, 0.6931471805599453);
  p = fma(p, f, 1.0);
  return ldexp(p, (int)n);
}
template <int KIND>
__device__ __forceinline__ double traj_shape(double q) {  // q = SCALE r^2 (may be slightly negative: dot-product form)
  if constexpr (KIND == KIND_RBF) {
    return traj_exp2(-fmax(q, 0.0));
  } else {
    constexpr double IL = 1.0 / TrajShape<KIND>::L2E;   // s = sqrt(c) r = u / log2(e)
    const double qc = fmax(q, TrajShape<KIND>::FLOOR);
    const double u = traj_sqrt(qc);
    if constexpr (KIND == KIND_M12) return traj_exp2(-u);
    else if constexpr (KIND == KIND_M32) return fma(IL, u, 1.0) * traj_exp2(-u);
    else return fma(IL * IL / 3.0, qc, fma(IL, u, 1.0)) * traj_exp2(-u);   // 1 + s + s^2 / 3
  }
}


// runtime-kind form for the cold kernels (assembly of K, K** blocks, predict_mean)
__device__ __forceinline__ double kernel_rt(int kind, double r2, double variance) {
  switch (kind) {
    case KIND_RBF: return kernel_from_r2<KIND_RBF>(r2, variance);
    case KIND_M12: return kernel_from_r2<KIND_M12>(r2, variance);
    case KIND_M32: return kernel_from_r2<KIND_M32>(r2, variance);
    default: return kernel_from_r2<KIND_M52>(r2, variance);
  }
}

// ---- normal distribution ------------------------------------------------------------------
__device__ __forceinline__ double normal_cdf(double z) {
  return 0.5 * erfc(-z * 0.7071067811865476);
}
__device__ __forceinline__ double normal_pdf(double z) {
  return 0.3989422804014327 * exp(-0.5 * z * z);
}

// d k / d (r^2) divided by the variance-free shape: returns variance * f'(r2).
__device__ __forceinline__ double kernel_dr2(int kind, double r2, double variance) {
  if (kind == KIND_RBF) return -0.5 * variance * exp(-0.5 * r2);
  const double r = sqrt(fmax(r2, 1e-36));
  if (kind == KIND_M12) return -0.5 * variance * exp(-r) / r;
  if (kind == KIND_M32) {
    const double s = 1.7320508075688772 * r;
    return -1.5 * variance * exp(-s);
  }
  const double s = 2.23606797749979 * r;
  return -(5.0 / 6.0) * variance * (1.0 + s) * exp(-s);
}

// Acquisition tails on (mean, clipped var)  -- reference function.py:220-223 (EI), 509-510 (PI),
// 415-416 (negative LCB), 319-325 (augmented EI: EI * (1 - sqrt(noise) / sqrt(noise + var))).
__device__ __forceinline__ double acq_tail(int acq, double param, double mean, double var, double noise) {
  const double sd = sqrt(var);
  if (acq == ACQ_EI || acq == ACQ_AEI) {
    const double diff = param - mean;
    const double z = diff / sd;
    const double ei = diff * normal_cdf(z) + sd * normal_pdf(z);
    if (acq == ACQ_EI) return ei;
    return ei * (1.0 - sqrt(noise) / sqrt(noise + var));
  } else if (acq == ACQ_PI) {
    return normal_cdf((param - mean) / sd);
  } else {
    return -(mean - param * sd);
  }
}

// ---- (value, index) ordering: larger value wins, ties -> smaller index (tf.math.argmax) -----
// NaN never wins (TF's argmax would propagate NaN; a NaN acquisition value is an upstream bug).
// log Phi(x), stable in both tails -- tfp.distributions.Normal.log_cdf (special_math.log_ndtr, float64):
// x > 8: -Phi(-x);  -20 <= x <= 8: log Phi(x);  x < -20: the asymptotic series of order 3
//   -x^2/2 - log(-x) - log(2 pi)/2 + log(1 - 1/x^2 + 3/x^4 - 15/x^6).
__device__ __forceinline__ double log_normal_cdf(double x) {
  if (x > 8.0) return -normal_cdf(-x);
  if (x >= -20.0) return log(normal_cdf(x));
  const double x2 = x * x;
  return -0.5 * x2 - log(-x) - 0.9189385332046727 + log(1.0 - 1.0 / x2 + 3.0 / (x2 * x2) - 15.0 / (x2 * x2 * x2));
}

// Entropy-search tails on (mean, var) and the min-value samples s_1..s_S (reference
// acquisition/function/entropy.py): with gamma_s = (s - mean) / sd, sd = max(sqrt(var), 1e-8) (CLAMP_LB :47),
// ratio = pdf(gamma) / Phi(-gamma) computed as exp(log pdf - log Phi(-gamma)):
//   ACQ_MES     (min_value_entropy_search.__call__ :195-214): mean_s [-gamma ratio / 2 - log Phi(-gamma)]
//   ACQ_GIBBON  (gibbon_quality_term.__call__ :479-500):      -1/2 mean_s log(1 + rho^2 ratio (gamma - ratio)),
//                                                              rho^2 = var / (var + noise)
//   ACQ_LOGYVAR: log(var + noise) -- the two halves of GIBBON's repulsion term (:580-619)
// Returns the value and its partial derivatives w.r.t. mean and var (for the L-BFGS-B refinement).
__device__ __forceinline__ void entropy_tail(int acq, double mean, double var, double noise,
                                             const double* __restrict__ samples, int S, double& v,
                                             double& dv_dmu, double& dv_dvar) {
  if (acq == ACQ_LOGYVAR) {
    v = log(var + noise);
    dv_dmu = 0.0;
    dv_dvar = 1.0 / (var + noise);
    return;
  }
  const double sd_raw = sqrt(var);
  const bool clamped = !(sd_raw > 1e-8);
  const double sd = clamped ? 1e-8 : sd_raw;
  const double rho2 = var / (var + noise);
  double acc = 0.0, acc_u = 0.0, acc_rho = 0.0, acc_uu = 0.0;
  for (int t = 0; t < S; ++t) {
    const double u = (samples[t] - mean) / sd;
    const double lmc = log_normal_cdf(-u);
    const double r = exp(-0.5 * u * u - 0.9189385332046727 - lmc);
    double f, fu;  // per-sample value and d/du;  dr/du = r (r - u)
    if (acq == ACQ_MES) {
      f = -0.5 * u * r - lmc;
      fu = 0.5 * r - 0.5 * u * r * (r - u);
    } else {
      const double hh = r * (u - r);
      const double inner = 1.0 + rho2 * hh;
      const double dh = -r * (u - r) * (u - r) + r - r * r * (r - u);
      f = -0.5 * log(inner);
      fu = -0.5 * rho2 * dh / inner;
      acc_rho += -0.5 * hh / inner;  // d/d rho^2
    }
    acc += f;
    acc_u += fu;
    acc_uu += fu * u;
  }
  const double inv = 1.0 / (double)S;
  v = acc * inv;
  // du/dmean = -1/sd;  du/dvar = -u / (2 var) (zero where sd is clamped)
  dv_dmu = -acc_u * inv / sd;
  dv_dvar = clamped ? 0.0 : -acc_uu * inv / (2.0 * var);
  if (acq == ACQ_GIBBON) dv_dvar += acc_rho * inv * noise / ((var + noise) * (var + noise));
}

__device__ __forceinline__ bool better(double v, int64_t i, double bv, int64_t bi) {
  return (v > bv) || (v == bv && i < bi);
}

__device__ __forceinline__ void wave_argmax(double& v, int64_t& i) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(v, off, 64);
    const int64_t oi = __shfl_xor((long long)i, off, 64);
    if (better(ov, oi, v, i)) {
      v = ov;
      i = oi;
    }
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- Philox4x32-10 -------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform double in [0,1) with 53 random bits, a pure function of (seed, element index)
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint64_t elem) {
  uint32_t o[4];
  philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), 0x7467705fu, 0u, (uint32_t)seed,
                (uint32_t)(seed >> 32), o);
  const uint64_t bits = ((uint64_t)o[0] << 32) | o[1];
  return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}

}  // namespace tgp
