// Host side of the C-ABI declared in include/tgp.h: handle state, device memory, the recursive
// Cholesky / triangular-inverse driver of `update`, staging of caller buffers and launch glue.
#include "../../include/tgp.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <tuple>
#include <new>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <vector>

#include "tgp_host.hpp"

namespace tgp {
hipError_t launch_sweep_kind0(hipStream_t, const SweepArgs&, bool, int64_t);
hipError_t launch_sweep_kind1(hipStream_t, const SweepArgs&, bool, int64_t);
hipError_t launch_sweep_kind2(hipStream_t, const SweepArgs&, bool, int64_t);
hipError_t launch_sweep_kind3(hipStream_t, const SweepArgs&, bool, int64_t);
hipError_t launch_sweep_dma_kind0(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_sweep_dma_kind1(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_sweep_dma_kind2(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_sweep_dma_kind3(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_joint_kind0(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_joint_kind1(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_joint_kind2(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_joint_kind3(hipStream_t, const SweepArgs&, int64_t);
hipError_t launch_sweep_i8_kind0(hipStream_t, const SweepArgs&, int64_t, int);
hipError_t launch_sweep_i8_kind1(hipStream_t, const SweepArgs&, int64_t, int);
hipError_t launch_sweep_i8_kind2(hipStream_t, const SweepArgs&, int64_t, int);
hipError_t launch_sweep_i8_kind3(hipStream_t, const SweepArgs&, int64_t, int);
}  // namespace tgp

using namespace tgp;

namespace {

thread_local std::string g_create_error;
std::atomic<uint64_t> g_data_version{0};  // stamps factorisations: unique across handles and threads

}  // namespace

namespace {

int fail(tgp_handle h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  else g_create_error = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(h, e_ == hipErrorOutOfMemory ? TGP_ERR_ALLOC : TGP_ERR_HIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                    \
  } while (0)

ModelDev model_dev(tgp_handle h) {
  ModelDev m;
  m.kind = h->kind;
  m.d = h->d;
  m.dp = h->dp;
  m.N = h->N;
  m.Npad = h->Npad;
  m.variance = h->variance;
  m.noise = h->noise;
  m.mean_const = h->mean_const;
  m.ls = h->d_ls.as<double>();
  m.Xs = h->d_Xs.as<double>();
  m.xn = h->d_xn.as<double>();
  m.Wt = h->d_A.as<double>();  // A is recycled as Wt after the factorisation
  m.alpha = h->d_alpha.as<double>();
  return m;
}

// Bring a caller array onto the device (no-op for TGP_DEVICE).
int stage_in(tgp_handle h, DevBuf& buf, const double* src, size_t count, int where,
             const double** out) {
  if (where == TGP_DEVICE) {
    *out = src;
    return TGP_OK;
  }
  HIPCHK(h, buf.reserve(count * sizeof(double)));
  HIPCHK(h, hipMemcpyAsync(buf.p, src, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  *out = buf.as<double>();
  return TGP_OK;
}

// Where a kernel should write an output of `count` doubles destined for dst.
int stage_out_prepare(tgp_handle h, DevBuf& buf, double* dst, size_t count, int where, double** dev) {
  if (!dst) {
    *dev = nullptr;
    return TGP_OK;
  }
  if (where == TGP_DEVICE) {
    *dev = dst;
    return TGP_OK;
  }
  HIPCHK(h, buf.reserve(count * sizeof(double)));
  *dev = buf.as<double>();
  return TGP_OK;
}

int stage_out_finish(tgp_handle h, const double* dev, double* dst, size_t count, int where) {
  if (!dst || where == TGP_DEVICE) return TGP_OK;
  HIPCHK(h, hipMemcpyAsync(dst, dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return TGP_OK;
}

int sync(tgp_handle h) {
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return TGP_OK;
}

int set_device(tgp_handle h) {
  HIPCHK(h, hipSetDevice(h->device));
  // hipGetLastError is per host thread and sticky: an error left behind by ANY earlier runtime call on
  // this thread (another library's, or a failed cleanup call of a garbage-collected handle) would be
  // reported by the next launch check of this call.  Every entry point starts from a clean slate.
  (void)hipGetLastError();
  h->auto_pinned = false;  // (a call that failed between sweep_blocks and its launch must not freeze the next call's choice)
  return TGP_OK;
}

// multiply acquisition values (device, [M]) by the handle's local penalization, if one is set
void apply_penalization(tgp_handle h, double* dvals, const double* dXq, int64_t M) {
  if (h->pen_kind == 0 || h->pen_P == 0 || !dvals) return;
  const double* pend = h->d_pen.as<double>();
  launch_penalize(h->stream, dvals, dXq, M, h->d, h->pen_kind, h->pen_P, pend, pend + (size_t)h->pen_P * h->d,
                  pend + (size_t)h->pen_P * (h->d + 1));
}

// Sweep launch policy.  tgp_set_variant bits (experiments / tests; 0 = default):
//   VARIANT_NO_SPLIT (1): never use the row-group split    VARIANT_FORCE_SPLIT (2): use it whenever Npad allows
//   VARIANT_JOINT_V1 (4): joint mode on the first-generation kernel (64-column slots, Gram operands from L2)
//   VARIANT_REG_STAGING (8): fused plain launches on the register-staged kernel instead of the LDS-DMA one
//   VARIANT_NO_DAG (16): `update` through the recursion of dependent launches instead of the persistent task-DAG kernel
//   VARIANT_DAG_SMALL (32): the persistent kernel from Npad = 256 on (default: from 512 on; rounds 3 - 5: 4096)
//   VARIANT_NO_REPAIR_PRODUCT (64): TGP_PREC_AUTO recomputes every flagged candidate through the SPLIT sweep (rounds 4 / 5)
//                                   instead of the product path for short lists
//   VARIANT_STATIC_BLOCKS (128): the int8 sweep's workgroups take candidate blocks i, i + #WG, ... instead of drawing them from a counter
//   VARIANT_DAG_WHOLE_TILES (256): the persistent `update` kernel's plan without the round-6 split of its two critical single products
//   VARIANT_DAG_ONE_CHAIN (512): the persistent `update` kernel's chain as ONE workgroup (rounds 3 - 5) instead of round 6's two
constexpr int VARIANT_NO_SPLIT = 1, VARIANT_FORCE_SPLIT = 2, VARIANT_JOINT_V1 = 4, VARIANT_REG_STAGING = 8, VARIANT_NO_DAG = 16,
              VARIANT_DAG_SMALL = 32, VARIANT_NO_REPAIR_PRODUCT = 64, VARIANT_STATIC_BLOCKS = 128, VARIANT_DAG_WHOLE_TILES = 256,
              VARIANT_DAG_ONE_CHAIN = 512, VARIANT_SWEEP_SMALL_PREDICT = 1024;
//   VARIANT_SWEEP_SMALL_PREDICT (1024): tgp_predict at <= 2048 points through a sweep launch (rounds 1 - 5) instead of the skinny product
constexpr int64_t REPAIR_PCAP = 512;   // TGP_PREC_AUTO: lists up to this many candidates are recomputed as a product
int gemm_tall(tgp_handle h, bool tb, int m, int n, int k, double alpha, const double* A, int64_t lda, const double* B,
              int64_t ldb, double beta, double* C, int64_t ldc, int tri);

// TGP_PREC_AUTO (round 4): the split-precision sweep WITH an a-posteriori repair (sweep_i8_repaired below) on a
// ladder four planes -> five planes (d <= 16) -> float64.  Every repaired sweep reports how many candidates it had to
// recompute in float64 (an 8-byte copy into pinned host memory, read back lazily -- no synchronisation); when more
// than AUTO_DEMOTE of a sweep's candidates were recomputed on the current rung the next sweep moves one rung down: from there
// on the recomputation costs more than the wider arithmetic saves.  Break-even with round 6's kernels at the headline size
// (us per candidate: four planes 0.0735, five 0.1067, float64 0.2527): 0.0735 + 0.2527 f = 0.1067 at f = 13 %; the
// threshold sits below it (rounds 4 / 5: 5 %, from 0.095 + 0.255 f = 0.125 with a margin of two).  tgp_set_hyper /
// tgp_set_precision restart the ladder at four planes.  Whatever the rung, every result is inside the parity tolerance as
// far as the 8-sigma bound of the error model goes (a statistical model of the dropped digit pairs, not a worst-case bound),
// and the fused arg-max returns the float64 winner.
constexpr double AUTO_DEMOTE = 0.10;
constexpr int64_t I8_MAX_N = 16384;  // int32 accumulators: 5 pairs x 2^14 x N < 2^31
int auto_rung_precision(tgp_handle h) {
  if (h->N > I8_MAX_N) return TGP_PREC_F64;
  if (h->auto_level == 0) return TGP_PREC_I8X4;
  if (h->auto_level == 1 && h->dp <= 16) return TGP_PREC_I8X5;
  return TGP_PREC_F64;
}
// the current epoch's canary counters (as last read) join the totals: the device's words restart with the epoch
static void auto_fold_counters(tgp_handle h) {
  for (int k = 0; k < 2; ++k) {
    h->can_checked_total[k] += h->can_checked[k];
    h->can_viol_total[k] += h->can_viol[k];
    h->can_checked[k] = h->can_viol[k] = 0;
  }
  h->can_slack_total += h->can_slack;
  h->can_slack = 0;
}
// the ladder starts over at four planes; `zero_report`: tgp_set_precision / tgp_set_auto_sigma also clear what
// tgp_get_auto_report accumulates (a restart because the hyper-parameters moved keeps it)
static void auto_restart(tgp_handle h, bool zero_report) {
  auto_fold_counters(h);
  h->auto_level = 0;
  ++h->auto_epoch;
  h->rep_last_M = h->rep_last_count = 0;
  if (zero_report) {
    for (int k = 0; k < 2; ++k) {
      h->can_checked_total[k] = h->can_viol_total[k] = 0;
      h->can_worst[k] = 0.0;
    }
    h->can_slack_total = 0;
    h->can_demotions = 0;
  }
  h->auto_hyp.clear();
  h->auto_hyper_dirty = false;
}
// the last COMPLETED repaired sweep's report (stream-ordered copy into pinned memory; a torn or stale read only delays the
// decision by one sweep): counters, and the decision to leave the rung
static void auto_read_report(tgp_handle h) {
  if (!h->rep_host) return;
  const volatile int64_t* r = h->rep_host;
  const int64_t cnt = r[RS_COUNT], M = r[RS_M], ep = r[RS_TAG];
  const int64_t viol[2] = {r[RS_VIOL], r[RS_ADV_VIOL]}, chk[2] = {r[RS_CHECKED], r[RS_ADV_CHECKED]};
  const int64_t wbits[2] = {r[RS_WORST], r[RS_ADV_WORST]};
  if (ep != (int64_t)h->auto_epoch || M <= 0) return;
  h->rep_last_count = cnt;
  h->rep_last_M = M;
  for (int k = 0; k < 2; ++k) {
    h->can_viol[k] = viol[k];
    h->can_checked[k] = chk[k];
    double worst;
    memcpy(&worst, &wbits[k], sizeof(double));
    if (worst == worst && worst > h->can_worst[k]) h->can_worst[k] = worst;
  }
  h->can_slack = r[RS_SLACK_SAVED];
  // a rung is left when the repair costs more than the wider arithmetic saves, or when a sampled candidate's float64
  // value lay outside the bound the int8 kernel priced it at: the error model is wrong for this input
  const bool costly = M >= 1024 && (double)cnt > AUTO_DEMOTE * (double)M;
  const bool violated = viol[0] + viol[1] > 0;
  if ((costly || violated) && h->auto_level < 2) {
    h->auto_level += (h->auto_level == 0 && h->dp > 16) ? 2 : 1;
    auto_fold_counters(h);
    ++h->auto_epoch;
    if (violated) ++h->can_demotions;
    h->auto_hyp.assign(1, h->variance / h->noise);
    h->auto_hyp.insert(h->auto_hyp.end(), h->ls.begin(), h->ls.end());
  }
}
hipError_t resolve_precision(tgp_handle h) {
  if (h->precision_req != TGP_PREC_AUTO) {
    h->precision = h->precision_req;
    h->repair = false;
    return hipSuccess;
  }
  if (h->auto_pinned) return hipSuccess;  // already resolved for the call in progress (sweep_blocks)
  auto_read_report(h);
  if (h->auto_hyper_dirty) {
    // tgp_set_hyper ran since the last sweep (it closed the epoch: reports of sweeps under the old hyper-parameters are
    // stale).  New hyper-parameters (new conditioning) restart the ladder -- unless a rung was left under hyper-parameters
    // within a factor two of the ones in effect NOW (variance / noise, every lengthscale): a BO loop whose refit moves them a
    // little would otherwise re-pay the failed rung's sweep plus its repair at every step.  Decided here, at the next sweep,
    // not inside tgp_set_hyper: the trial evaluations of a fit (prior draws, L-BFGS-B steps) pass through far-away values
    // and come back.
    h->auto_hyper_dirty = false;
    bool keep = h->auto_level > 0 && h->auto_hyp.size() == (size_t)h->d + 1;
    if (keep) {
      auto far = [](double a, double b) { return !(a < 2.0 * b && b < 2.0 * a); };
      keep = !far(h->variance / h->noise, h->auto_hyp[0]);
      for (int c = 0; keep && c < h->d; ++c) keep = !far(h->ls[c], h->auto_hyp[1 + c]);
    }
    if (!keep) auto_restart(h, false);
  }
  h->precision = auto_rung_precision(h);
  h->repair = h->precision != TGP_PREC_F64;
  return hipSuccess;
}
// after a synchronising call under TGP_PREC_AUTO: did a canary of the sweeps just completed fire?  Then the rung is left
// NOW and the caller repeats its sweeps, so that what it returns was computed on a rung whose samples all held
static bool auto_canary_tripped(tgp_handle h) {
  if (h->precision_req != TGP_PREC_AUTO || !h->repair || !h->rep_host || h->auto_level >= 2) return false;
  const volatile int64_t* r = h->rep_host;
  if (r[RS_TAG] != (int64_t)h->auto_epoch || r[RS_VIOL] + r[RS_ADV_VIOL] <= 0) return false;
  h->auto_pinned = false;
  (void)resolve_precision(h);
  return true;
}

// number of per-block winner slots a fused arg-max over `a` fills (one per candidate block of the kernel in use)
int64_t sweep_blocks(tgp_handle h, const SweepArgs& a, bool joint) {
  if (!joint) {
    (void)resolve_precision(h);
    h->auto_pinned = true;  // launch_sweep_timed of the same call must see the same choice
  }
  if (!joint && h->precision != TGP_PREC_F64) return (a.M + 63) / 64;
  return sweep_grid(a, joint);
}

// row-group split of the f64 sweep: groups of row blocks of roughly equal triangular work
bool plan_split(SweepArgs& am, int nb, int g) {
  const int total = nb * (nb + 1) / 2;
  am.split_ib[0] = 0;
  int ib = 0;
  for (int k = 1; k < g; ++k) {
    while (ib < nb && ib * (ib + 1) / 2 < (int64_t)total * k / g) ++ib;
    am.split_ib[k] = std::max(ib, am.split_ib[k - 1] + 1);
  }
  am.split_ib[g] = nb;
  bool ok = true;
  for (int k = 0; k < g; ++k) ok = ok && am.split_ib[k] < am.split_ib[k + 1];
  return ok;
}

hipError_t launch_sweep_kind(tgp_handle h, const SweepArgs& a, bool joint, int64_t wgrid) {
  switch (h->kind) {
    case TGP_RBF: return launch_sweep_kind0(h->stream, a, joint, wgrid);
    case TGP_MATERN12: return launch_sweep_kind1(h->stream, a, joint, wgrid);
    case TGP_MATERN32: return launch_sweep_kind2(h->stream, a, joint, wgrid);
    default: return launch_sweep_kind3(h->stream, a, joint, wgrid);
  }
}

// split-precision sweep: digit planes of W (once per factorisation), 64-candidate blocks, 4 B / entry K* slabs
hipError_t launch_sweep_i8_timed(tgp_handle h, SweepArgs& am) {
  const int64_t Npad = am.m.Npad;
  hipError_t e;
  const int planes = h->precision == TGP_PREC_I8X5 ? 5 : 4;
  if (h->wq_version != h->data_version || h->wq_planes != planes) {
    if ((e = h->d_wq.reserve((size_t)planes * Npad * Npad)) != hipSuccess) return e;
    if ((e = h->d_rs.reserve((size_t)2 * Npad * sizeof(double))) != hipSuccess) return e;
    launch_w_digits(h->stream, h->d_W.as<double>(), h->N, Npad, h->d_rs.as<double>(), h->d_wq.p, planes);
    if (h->dp <= 16) {   // the generating steps' training rows as DMA-able tiles
      const int xt = i8_xs_tile_doubles(h->dp);
      if ((e = h->d_xsa.reserve((size_t)(Npad / 32) * xt * sizeof(double))) != hipSuccess) return e;
      launch_xs_tiles(h->stream, h->d_Xs.as<double>(), h->d_alpha.as<double>(), Npad, h->dp, xt, h->d_xsa.as<double>());
    }
    h->wq_version = h->data_version;
    h->wq_planes = planes;
  }
  am.i8_wq = h->d_wq.p;
  am.i8_rs = h->d_rs.as<double>();
  am.i8_xsa = i8_tiles_fit(planes, h->dp) ? h->d_xsa.as<double>() : nullptr;   // (five planes: two tile buffers, d <= 8)
  const int64_t blocks = (am.M + 63) / 64;
  const int64_t wgrid = blocks < h->num_cu ? blocks : h->num_cu;
  // candidate blocks beyond one per workgroup are drawn from a counter (the workgroups of a launch differ in speed)
  am.blk_ctr = nullptr;
  if (blocks > wgrid && !(h->variant & VARIANT_STATIC_BLOCKS)) {
    if ((e = h->s_blkctr.reserve(64)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(h->s_blkctr.p, 0, sizeof(unsigned), h->stream)) != hipSuccess) return e;
    am.blk_ctr = h->s_blkctr.as<unsigned>();
  }
  if (!h->repair) {
    if ((e = h->s_kcache.reserve((size_t)wgrid * (size_t)Npad * 64 * planes)) != hipSuccess) return e;
    am.kcache = h->s_kcache.as<double>();
    am.rep_ub = nullptr;
    am.canary_rec = nullptr;
    am.adv_rec = nullptr;
    (void)hipEventRecord(h->ev0, h->stream);
    switch (h->kind) {
      case TGP_RBF: e = launch_sweep_i8_kind0(h->stream, am, wgrid, planes); break;
      case TGP_MATERN12: e = launch_sweep_i8_kind1(h->stream, am, wgrid, planes); break;
      case TGP_MATERN32: e = launch_sweep_i8_kind2(h->stream, am, wgrid, planes); break;
      default: e = launch_sweep_i8_kind3(h->stream, am, wgrid, planes); break;
    }
    (void)hipEventRecord(h->ev1, h->stream);
    h->last_launches = 1;
    h->last_ms = -1.0;
    return e;
  }
  // ---- with the a-posteriori repair (TGP_PREC_AUTO) -------------------------------------------------------------
  //  1 int8 sweep: values, per-candidate intervals (rep_ub; +inf = outside the parity tolerance), block winners = the
  //    largest LOWER interval ends                         2 L = the largest lower end of the sweep
  //  3 list = {ub == +inf} u {ub >= L} (whatever violates the tolerance, whatever could still be the f64 arg-max)
  //  4 gather -> float64 sweep (SPLIT instantiation, the count stays on the device) -> scatter over the int8 results
  //  5 block winners := per-slot arg-max of the patched values.     Nothing here synchronises with the host.
  const int64_t M = am.M;
  const int d = am.m.d;
  const int nb = (int)(Npad / NPAD_MULT);
  SweepArgs b{};
  // one row block per group where there are at most 16: a handful of recomputed candidates is ONE candidate block, and
  // its latency is the heaviest group's walk (N = 4096: the last row block alone is 12 % of the triangle, 0.9 ms)
  int g = std::min(16, nb);
  if (g == nb) {
    for (int k = 0; k <= nb; ++k) b.split_ib[k] = k;
  } else {
    while (g > 1 && !plan_split(b, nb, g)) --g;  // (not every group count leaves every group a row block)
  }
  if (g == 1) {
    b.split_ib[0] = 0;
    b.split_ib[1] = nb;
  }
  const int64_t cap = M + I8_ADV_GROUPS;   // the repair list: at most every candidate + the adversarial picks
  const int64_t cap_blocks = (cap + SW_BN - 1) / SW_BN;
  int64_t fgrid = cap_blocks * g;
  fgrid = fgrid < h->num_cu ? fgrid : h->num_cu;
  const size_t kc_i8 = (size_t)wgrid * (size_t)Npad * 64 * planes, kc_f64 = (size_t)fgrid * (size_t)Npad * SW_BN * sizeof(double);
  if ((e = h->s_kcache.reserve(std::max(kc_i8, kc_f64))) != hipSuccess) return e;
  if ((e = h->s_part.reserve((size_t)cap_blocks * g * 256 * sizeof(double))) != hipSuccess) return e;
  // ub [M], vals [M]; with room for the adversarial picks behind a list of everything (cap = M + 64): recomputed (mean, var,
  // acq) [3][cap], list [cap] (int64), gathered candidates [cap][d]; then the uniform canary's records [n_can][2], the
  // int8 sweep's per-block adversarial records [blocks][4] and the picks [64][3]
  const int64_t n_can = (M + I8_CANARY_PERIOD - 1) / I8_CANARY_PERIOD + 2;
  if ((e = h->s_rep.reserve((size_t)(2 * M + cap * (4 + d) + 2 * n_can + 4 * blocks + 3 * I8_ADV_GROUPS) * sizeof(double) + 64)) !=
      hipSuccess)
    return e;
  if ((e = h->s_rep_stats.reserve(RS_WORDS * sizeof(int64_t))) != hipSuccess) return e;
  if (!h->rep_host && (e = hipHostMalloc((void**)&h->rep_host, RS_WORDS * sizeof(int64_t), hipHostMallocDefault)) != hipSuccess)
    return e;
  double* ub = h->s_rep.as<double>();
  double* vals = ub + M;          // acquisition values when the caller wants none written
  double* rout = vals + M;        // [3][cap] mean, var, acq of the recomputed candidates
  int64_t* list = (int64_t*)(rout + 3 * cap);
  double* Xg = (double*)(list + cap);
  double* crec = Xg + cap * d;                      // [n_can][2]: the sampled candidates' int8 variance and bound
  double* adv_rec = crec + 2 * n_can;               // [blocks][4]
  double* adv_sel = adv_rec + 4 * blocks;           // [64][3]
  int64_t* stats = h->s_rep_stats.as<int64_t>();   // tgp_internal.hpp RS_*
  double* Lslot = (double*)(stats + RS_L);          // {L, its index}
  int64_t* route = stats + RS_ROUTE;                // {count if the SPLIT sweep recomputes, count if the product path does}
  // Up to pcap recomputed candidates (the canary's M / 4096 and a handful of flagged ones: the usual case) take the
  // product path of tgp_kernels_misc.hip -- K*^T, W K*^T as a tall product, column sums, tail -- which spreads over the chip;
  // more than that go through the SPLIT sweep.  Both are enqueued, repair_route_kernel gives the count to one of them.
  // pcap follows the list the sweep is expected to leave -- the two strata of the canary plus a margin for the arg-max band
  // (the product's K*^T / gemm / column sums run whatever the count is, so small sweeps get a small product)
  const int64_t pcap = std::min<int64_t>(REPAIR_PCAP, ((n_can + std::min<int64_t>(I8_ADV_GROUPS, blocks) + 128 + 63) / 64) * 64);
  const bool product = !(h->variant & VARIANT_NO_REPAIR_PRODUCT);
  double *pB = nullptr, *pC = nullptr, *ppart = nullptr;
  if (product) {
    if ((e = h->s_grad.reserve(((size_t)2 * Npad * pcap + (size_t)repair_product_part_doubles(pcap)) * sizeof(double))) != hipSuccess)
      return e;
    pB = h->s_grad.as<double>();
    pC = pB + (size_t)Npad * pcap;
    ppart = pC + (size_t)Npad * pcap;
  }
  // the uniform sample of this sweep: candidates j with (j + off) % 4096 == 0, off = a hash of (N, hyper-parameters, M, rung)
  // -- a function of the model and the call's shape, not of a call counter (round 5) nor of the handle (two handles on the
  // same model sample alike): two identical calls sample, decide and return the same
  auto mix = [](uint64_t zz, uint64_t x) {
    zz = (zz ^ x) * 0xBF58476D1CE4E5B9ull;
    return zz ^ (zz >> 29);
  };
  auto bits = [](double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return u;
  };
  uint64_t z = mix(0x9E3779B97F4A7C15ull, (uint64_t)h->N);
  z = mix(z, (uint64_t)M);
  z = mix(z, (uint64_t)h->auto_level);
  z = mix(z, bits(h->variance));
  z = mix(z, bits(h->noise));
  for (int c = 0; c < h->d; ++c) z = mix(z, bits(h->ls[c]));
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  const int64_t can_off = (int64_t)((z ^ (z >> 31)) & (uint64_t)(I8_CANARY_PERIOD - 1));
  am.canary_rec = crec;
  am.canary_off = can_off;
  am.adv_rec = adv_rec;
  const double v = h->variance, eps = 2.220446049250313e-16;
  am.rep_ub = ub;
  am.rep_floor = std::min(64.0 * eps * v * (1.0 + (double)h->N * v / h->noise), 1e-6 * v);
  // K_SIGMA = 8 standard deviations of the error model (tools/ozaki_tight.py: observed / model rms = 0.95 ... 0.97,
  // max over 2048 candidates 3.5); dropped digit pairs: 3 at weight 2^-32 (four planes), 4 at 2^-40 (five)
  am.rep_scale = h->auto_sigma * 2.0 * (planes == 4 ? std::exp2(-32.8) : std::exp2(-40.6)) * (I8_TIGHT * v);
  double* user_acq = am.acq_out;
  double* ublk_val = am.blk_val;
  int64_t* ublk_idx = am.blk_idx;
  if (am.acq_kind >= 0 && !am.acq_out) am.acq_out = vals;
  am.kcache = h->s_kcache.as<double>();
  (void)hipEventRecord(h->ev0, h->stream);
  launch_repair_begin(h->stream, stats, M, (int64_t)h->auto_epoch, h->canary_epoch != h->auto_epoch);
  h->canary_epoch = h->auto_epoch;
  switch (h->kind) {
    case TGP_RBF: e = launch_sweep_i8_kind0(h->stream, am, wgrid, planes); break;
    case TGP_MATERN12: e = launch_sweep_i8_kind1(h->stream, am, wgrid, planes); break;
    case TGP_MATERN32: e = launch_sweep_i8_kind2(h->stream, am, wgrid, planes); break;
    default: e = launch_sweep_i8_kind3(h->stream, am, wgrid, planes); break;
  }
  if (e != hipSuccess) return e;
  if (ublk_val) launch_argmax_final(h->stream, ublk_val, ublk_idx, blocks, Lslot, (int64_t*)(Lslot + 1));
  launch_repair_flag(h->stream, ub, M, ublk_val ? Lslot : nullptr, list, stats, can_off);
  launch_repair_adv(h->stream, adv_rec, blocks, list, stats, adv_sel);
  launch_repair_gather(h->stream, am.Xq, d, list, stats, cap, Xg);
  b.m = am.m;
  b.Xq = Xg;
  b.M = cap;
  b.M_dev = stats;
  if (product) {
    launch_repair_route(h->stream, stats, pcap, route);
    b.M_dev = route;   // zero unless the list is longer than pcap
  }
  b.mean_out = am.mean_out ? rout : nullptr;
  b.var_out = rout + cap;   // (always: the canary compares the recomputed variances)
  b.acq_out = am.acq_kind >= 0 ? rout + 2 * cap : nullptr;
  b.acq_kind = am.acq_kind;
  b.acq_param = am.acq_param;
  b.split_g = g;
  b.part = h->s_part.as<double>();
  b.kcache = h->s_kcache.as<double>();
  if ((e = launch_sweep_kind(h, b, false, fgrid)) != hipSuccess) return e;
  launch_sweep_combine(h->stream, b, std::min<int64_t>(cap_blocks, 2048));
  if (product) {
    launch_kstar_t_dev(h->stream, am.m, Xg, route + 1, pcap, pB);
    if (gemm_tall(h, false, (int)Npad, (int)pcap, (int)Npad, 1.0, h->d_W.as<double>(), Npad, pB, pcap, 0.0, pC, pcap, 3) != TGP_OK)
      return hipErrorOutOfMemory;
    launch_repair_product_tail(h->stream, am.m, pB, pC, pcap, route + 1, b.acq_kind, b.acq_param, ppart, b.mean_out, b.var_out,
                               b.acq_out);
  }
  // float64 against int8 on the sampled candidates.  Slack: the float64 reference's own rounding -- the two recomputation
  // paths sum in different orders, and var = s_f^2 - |W k*|^2 is a difference whose rounding grows with N and, through
  // |W| ~ 1 / s, with 1 / sqrt(noise): 4 eps N s_f^2 sqrt(s_f^2 / s^2), never below 1e-12 s_f^2 (round 5's fixed value) and
  // never above the parity tolerance's absolute floor (that floor itself, 1e-6 s_f^2 at low noise, would hide
  // real int8 errors: measured worst |d var| 2.6e-8 on the N = 1000, s^2 = 1e-5 model with the bound made 400 x too tight)
  const double can_slack = std::min(am.rep_floor, std::max(1e-12 * v, 4.0 * eps * (double)h->N * v * std::sqrt(v / h->noise)));
  launch_repair_canary(h->stream, list, stats, cap, b.var_out, crec, can_off, can_slack, adv_sel);
  launch_repair_scatter(h->stream, list, stats, cap, b.mean_out, am.var_out ? b.var_out : nullptr, b.acq_out, am.mean_out,
                        am.var_out, am.acq_out, ub, ublk_val ? Lslot : nullptr);
  if (ublk_val) launch_values_argmax(h->stream, am.acq_out, M, am.index_base, ublk_val, ublk_idx, blocks);
  (void)hipEventRecord(h->ev1, h->stream);
  e = hipMemcpyAsync(h->rep_host, stats, RS_WORDS * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream);
  am.acq_out = user_acq;
  h->last_launches = 1;
  h->last_ms = -1.0;
  return e;
}

hipError_t launch_sweep_timed(tgp_handle h, const SweepArgs& a, bool joint) {
  const int64_t grid = sweep_grid(a, joint);
  if (grid <= 0) return hipSuccess;
  hipError_t e;
  SweepArgs& am = const_cast<SweepArgs&>(a);
  am.split_g = 0;
  if (!joint) {
    e = resolve_precision(h);
    h->auto_pinned = false;
    if (e != hipSuccess) return e;
    // the digit planes, row scales and training-row tiles are built from THIS handle's factorisation (and cached per its
    // data_version): a sweep over another handle's model -- the repulsion twin of the entropy tails -- runs in float64
    const bool own_model = a.m.alpha == h->d_alpha.as<double>() && a.m.Npad == h->Npad;
    if (h->precision != TGP_PREC_F64 && own_model) return launch_sweep_i8_timed(h, am);
  }
  if (joint && a.m.dp <= 16 && !(h->variant & VARIANT_JOINT_V1)) {
    // contiguously packed 128 x 256 tiles, Gram phase out of LDS (tgp_kernels_joint.inc)
    const int gpb = 256 / a.q;
    const int64_t blocks = (a.G + gpb - 1) / gpb;
    const int64_t wg = blocks < h->num_cu ? blocks : h->num_cu;
    const size_t slab = (size_t)wg * (size_t)a.m.Npad * 256 * sizeof(double);
    hipError_t ea = h->s_kcache.reserve(slab);
    if (ea == hipSuccess) ea = h->s_aslab.reserve(slab);
    if (ea != hipSuccess) return ea;
    am.kcache = h->s_kcache.as<double>();
    am.aslab = h->s_aslab.as<double>();
    // (candidate blocks stay dealt, i, i + #WG, ...: drawn from a counter as in the int8 sweep the draw itself gains 0.5 % at
    // C4's 10 000 blocks, but the kernel compiled with it runs 1.7 % slower in either mode -- profiles/r06_joint_dynblocks.txt)
    (void)hipEventRecord(h->ev0, h->stream);
    switch (h->kind) {
      case TGP_RBF: e = launch_joint_kind0(h->stream, a, wg); break;
      case TGP_MATERN12: e = launch_joint_kind1(h->stream, a, wg); break;
      case TGP_MATERN32: e = launch_joint_kind2(h->stream, a, wg); break;
      default: e = launch_joint_kind3(h->stream, a, wg); break;
    }
    (void)hipEventRecord(h->ev1, h->stream);
    h->last_launches = 1;
    h->last_ms = -1.0;
    return e;
  }
  const bool want_split = (h->variant & VARIANT_FORCE_SPLIT) ||
                          (grid < 4 * (int64_t)h->num_cu && !(h->variant & VARIANT_NO_SPLIT));
  if (!joint && want_split) {
    // Few candidate blocks (EGO's default sweep is max(5000, 1000 d) candidates = 63 blocks at d = 8): one
    // workgroup per block walks all of W alone (7.5 ms at N = 4096) on a quarter of the CUs.  Split every
    // block's row blocks of W into up to 8 groups of roughly equal triangular work.
    const int nb = (int)(a.m.Npad / 256);
    int g = (int)std::min<int64_t>(std::min(8, nb), (8 * (int64_t)h->num_cu + grid - 1) / grid);
    if (h->variant & VARIANT_FORCE_SPLIT) g = std::min(8, nb);
    // (round 6: with as many groups as row blocks the equal-work boundaries leave a group empty -- nb = 4 and nb = 8, i.e.
    // N <= 1024 and N <= 2048, never split at all and EGO's 8000-candidate sweep ran on 125 of 256 compute units: 2.2 ms at
    // N = 2048 for 0.5 ms of arithmetic; take the largest group count that works)
    while (g > 1 && !plan_split(am, nb, g)) --g;
    if (g > 1) {
      const bool ok = true;
      if (ok) {
        am.split_g = g;
        hipError_t ep = h->s_part.reserve((size_t)grid * g * 256 * sizeof(double));
        if (ep != hipSuccess) return ep;
        am.part = h->s_part.as<double>();
      }
    }
  }
  if (!joint && am.split_g <= 1 && a.m.dp <= 16 && !(h->variant & VARIANT_REG_STAGING)) {
    // fused plain launch: LDS-DMA staging, three A stages (tgp_kernels_sweep_dma.inc); same results bit for bit
    const int64_t wg = grid < h->num_cu ? grid : h->num_cu;
    hipError_t ea = h->s_kcache.reserve((size_t)wg * (size_t)a.m.Npad * SW_BN * sizeof(double));
    if (ea != hipSuccess) return ea;
    am.kcache = h->s_kcache.as<double>();
    (void)hipEventRecord(h->ev0, h->stream);
    switch (h->kind) {
      case TGP_RBF: e = launch_sweep_dma_kind0(h->stream, a, wg); break;
      case TGP_MATERN12: e = launch_sweep_dma_kind1(h->stream, a, wg); break;
      case TGP_MATERN32: e = launch_sweep_dma_kind2(h->stream, a, wg); break;
      default: e = launch_sweep_dma_kind3(h->stream, a, wg); break;
    }
    (void)hipEventRecord(h->ev1, h->stream);
    h->last_launches = 1;
    h->last_ms = -1.0;
    return e;
  }
  // persistent: one workgroup per CU, each with a private K* slab (and a C slab in joint mode)
  int64_t wgrid = grid * std::max(1, am.split_g);
  wgrid = wgrid < h->num_cu ? wgrid : h->num_cu;
  hipError_t ea = h->s_kcache.reserve((size_t)wgrid * (size_t)a.m.Npad * SW_BN * sizeof(double));
  if (ea != hipSuccess) return ea;
  am.kcache = h->s_kcache.as<double>();
  if (joint) {
    ea = h->s_aslab.reserve((size_t)wgrid * (size_t)a.m.Npad * SW_BN * sizeof(double));
    if (ea != hipSuccess) return ea;
    am.aslab = h->s_aslab.as<double>();
  }
  (void)hipEventRecord(h->ev0, h->stream);
  e = launch_sweep_kind(h, a, joint, wgrid);
  if (e == hipSuccess && am.split_g > 1) launch_sweep_combine(h->stream, a, grid);
  (void)hipEventRecord(h->ev1, h->stream);
  h->last_launches = 1;
  h->last_ms = -1.0;  // resolved lazily in tgp_last_kernel_ms
  return e;
}

// ---- recursive Cholesky + inverse:  A (SPD, lower used) -> L, W = L^-1, all ld = Npad ----------
//   chol_inv(lo, hi):  leaf (64):  L_dd, W_dd from A_dd
//     else  chol_inv(lo, mid);  L21 = A21 W11^T;  A22 -= L21 L21^T;  chol_inv(mid, hi);
//           T = L21 W11 (into the dead A21);  W21 = -W22 T.
struct FactorWs {  // a square workspace: A (in, destroyed), L, W (out), leading dimension, info flag
  double *A, *L, *W;
  int64_t ld;
  int* info;
};

void chol_inv(hipStream_t st, const FactorWs& f, int64_t lo, int64_t hi) {
  const int64_t ld = f.ld;
  double *A = f.A, *L = f.L, *W = f.W;
  const int64_t n = hi - lo;
  if (n <= LEAF) {
    launch_leaf(st, A, L, W, ld, lo, f.info);
    return;
  }
  static const bool leaf128 = getenv("TGP_NO_LEAF128") == nullptr;  // A/B aid
  if (n == 2 * LEAF && leaf128) {  // two leaves and their parent's products in one workgroup
    launch_leaf128(st, A, L, W, ld, lo, f.info);
    return;
  }
  const int64_t nblk = n / LEAF;
  const int64_t mid = lo + (nblk / 2) * LEAF;
  const int s1 = (int)(mid - lo), s2 = (int)(hi - mid);
  chol_inv(st, f, lo, mid);
  double* A21 = A + mid * ld + lo;
  double* L21 = L + mid * ld + lo;
  double* W11 = W + lo * ld + lo;
  double* A22 = A + mid * ld + mid;
  launch_gemm(st, true, s2, s1, s1, 1.0, A21, ld, W11, ld, 0.0, L21, ld, false, 1);
  // A22 -= L21 L21^T and T = L21 W11 (into the dead A21) are independent: they share one launch instead of queueing
  // behind each other -- at every node size (the two grids fill each other's last, partly empty round of workgroups:
  // N = 4096 update 2.79 -> 2.64 ms when the nodes >= 1024 joined in; running T on a CU-masked side stream underneath
  // the right half's factorisation instead measured 2.72 ms)
  static const bool pair = getenv("TGP_NO_PAIR") == nullptr;  // A/B aid
  const bool fused = pair;
  if (fused) {
    launch_node_pair(st, s2, s1, L21, A22, W11, A21, ld);
  } else {
    launch_gemm(st, true, s2, s2, s1, -1.0, L21, ld, L21, ld, 1.0, A22, ld, true);
  }
  chol_inv(st, f, mid, hi);
  double* W22 = W + mid * ld + mid;
  double* W21 = W + mid * ld + lo;
  if (!fused) launch_gemm(st, false, s2, s1, s1, 1.0, L21, ld, W11, ld, 0.0, A21, ld, false, 2);
  launch_gemm(st, false, s2, s1, s2, -1.0, W22, ld, A21, ld, 0.0, W21, ld, false, 3);
}

void chol_inv(tgp_handle h, int64_t lo, int64_t hi) {
  chol_inv(h->stream, FactorWs{h->d_A.as<double>(), h->d_L.as<double>(), h->d_W.as<double>(), h->Npad,
                               h->d_info.as<int>()}, lo, hi);
}

// The whole factorisation + inverse as ONE persistent launch (tgp_kernels_dag.hip) for 512 <= Npad <= 16128 (rounds 3 - 5: from
// 4096 on -- 0.33 / 0.57 / 1.07 ms against the recursion's 0.24 / 0.45 / 0.98 ms at N = 512 / 1024 / 2048 THEN; tgp_set_variant
// bit 5 lowers the bound to 256 -- the tests use it; tile
// offsets in bytes fit 31 bits); the recursion above stays for everything else (small blocks, the append path, the
// q x q / F x F factorisations of the samplers).  TGP_NO_DAG=1 forces the recursion (A/B aid, tests).
bool dag_applies(tgp_handle h, int64_t Npad) {
  static const bool off = getenv("TGP_NO_DAG") != nullptr;
  // Npad >= 512 since round 6 (rounds 3 - 5: 4096).  Measured with the two-workgroup chain (profiles/r06_dag_small_sizes*.txt): `update`
  // 0.84 against the recursion's 0.99 ms at N = 2048, 1.28 / 2.00 at 3072, 1.62 / 2.42 at 3840, equal at 900 - 1024, 0.28 / 0.24 at
  // <= 512 -- and, what decides it, the hyper-parameter fit's prior draws go through tgp_nlml_trial_batch wherever this says yes:
  // find_best_model_initialization(90) 12 -> 4 ms at N = 512, 30 -> 7 at 1024, 55 -> 16 at 2048, 102 -> 34 at 3072; a cold
  // optimize() 21 -> 12, 35 -> 20, 67 -> 37, 128 -> 65 ms.  TGP_DAG_MIN_N overrides the bound (development aid).
  static const int64_t min_env = getenv("TGP_DAG_MIN_N") ? atoll(getenv("TGP_DAG_MIN_N")) : 512;
  const int64_t min_n = (h->variant & VARIANT_DAG_SMALL) ? 256 : min_env;
  return !off && !(h->variant & VARIANT_NO_DAG) && Npad >= min_n && Npad % 128 == 0 && Npad * Npad * 8 < (int64_t)0x7fffffff;
}

// The plan of `slot` (0 full update, 1 factor-only, >= 2 batched factor-only) for this size / share of the compute units /
// batch.  Plans are immutable and depend on nothing but (kind, NB, ld, workgroups, B): they live in a PROCESS-WIDE cache
// per device, built on the first miss (host simulation 2 - 10 ms, three copies, one synchronisation) -- a BO loop fits
// a fresh or re-attached model every step, and each used to rebuild its three plans (~15 ms of a 115 ms fit).
struct SharedPlan {
  DevBuf tasks, chain, topo;
  int ntasks = 0;
};
// the split plan (dag_build split_critical): the single full update where its chain is the bound
static bool dag_split(tgp_handle h, int slot, int NB) {
  static const int max_nb = getenv("TGP_DAG_SPLIT_MAX_NB") ? atoi(getenv("TGP_DAG_SPLIT_MAX_NB")) : 48;   // (development aid)
  return slot == 0 && NB >= 3 && NB < max_nb && !(h->variant & VARIANT_DAG_WHOLE_TILES);
}
// the two-workgroup chain (tgp_kernels_dag.hip run_duo): the single full update, wherever the split plan applies
static bool dag_duo(tgp_handle h, int slot, int NB) {
  return dag_split(h, slot, NB) && !(h->variant & VARIANT_DAG_ONE_CHAIN);
}
int dag_plan_get(tgp_handle h, int slot, int NB, int64_t ld, int grid, int B) {
  tgp_handle_s::DagPlan& p = h->dag_plan[slot];
  const bool split = dag_split(h, slot, NB), duo = dag_duo(h, slot, NB);
  if (p.nb == NB && p.ld == ld && p.grid == grid && p.B == B && p.split == split && p.duo == duo) return TGP_OK;
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int64_t, int, int>, SharedPlan*> cache;  // (never freed: process lifetime)
  std::lock_guard<std::mutex> lk(mu);
  const int kind = slot == 0 ? (duo ? 4 : (split ? 3 : 0)) : (slot == 1 ? 1 : 2);
  SharedPlan*& sp = cache[std::make_tuple(h->device, kind, NB, ld, grid, B)];
  if (!sp) {
    std::vector<DagTask> tasks;
    std::vector<uint32_t> chain, topo, merged;
    int nu = 0;
    // the dispatch order is simulated for the workers there are: alone, or B members sharing grid - B of them
    dag_build(NB, ld, tasks, chain, nu, &topo, std::max(1, slot >= 2 ? (grid - B) / B : grid - (duo ? 2 : 1)), slot == 0, slot >= 2 ? B : 1,
              grid - B, &merged, split, duo);
    if (slot >= 2) {
      if (B == 1) dag_merge_order(topo, 1, merged);
      topo.swap(merged);
    }
    SharedPlan* fresh = new SharedPlan();
    hipError_t e = fresh->tasks.reserve(tasks.size() * sizeof(DagTask));
    if (e == hipSuccess) e = fresh->chain.reserve(chain.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = fresh->topo.reserve((topo.size() + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(fresh->tasks.p, tasks.data(), tasks.size() * sizeof(DagTask), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(fresh->chain.p, chain.data(), chain.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(fresh->topo.p, topo.data(), topo.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // the host vectors die here; the plan is complete for every stream
    if (e != hipSuccess) {
      fresh->tasks.release();
      fresh->chain.release();
      fresh->topo.release();
      delete fresh;
      cache.erase(std::make_tuple(h->device, kind, NB, ld, grid, B));
      HIPCHK(h, e);
    }
    fresh->ntasks = (int)tasks.size();
    sp = fresh;
  }
  p.nb = NB;
  p.ld = ld;
  p.grid = grid;
  p.B = B;
  p.split = split;
  p.duo = duo;
  p.ntasks = sp->ntasks;
  p.tasks = sp->tasks.p;
  p.chain = sp->chain.p;
  p.topo = sp->topo.p;
  return TGP_OK;
}

// -> TGP_OK / error; the launch's own error words are read back by dag_check after the stream has drained
int chol_inv_dag(tgp_handle h, bool factor_only = false) {
  const int64_t Npad = h->Npad;
  const int NB = (int)(Npad / 128);
  // workgroups of the launch: all compute units, or this handle's share of them (tgp_set_update_concurrency)
  const int grid = std::max(std::min(32, h->num_cu), h->num_cu / std::max(1, h->update_share));
  const int slot = factor_only ? 1 : 0;
  if (int rc = dag_plan_get(h, slot, NB, Npad, grid, 1)) return rc;
  const tgp_handle_s::DagPlan& plan = h->dag_plan[slot];
  h->dag_last_slot = slot;
  h->dag_state_words = (size_t)plan.ntasks + 2 * (size_t)NB + DAG_CTRL_WORDS + (size_t)plan.ntasks + (plan.duo ? (size_t)DAG_DUO_PF * (size_t)NB : 0);
  HIPCHK(h, h->d_dag_flags.reserve(h->dag_state_words * sizeof(uint32_t)));
  const size_t nflags = (size_t)plan.ntasks + 2 * (size_t)NB;
  // flags, control words and start counts all start from zero, before EVERY launch
  HIPCHK(h, hipMemsetAsync(h->d_dag_flags.p, 0, h->dag_state_words * sizeof(uint32_t), h->stream));
  DagArgs a{};
  a.Ap = h->d_A.as<double>();
  a.Lp = h->d_L.as<double>();
  a.Wp = h->d_W.as<double>();
  a.ld = Npad;
  a.NB = NB;
  a.ntasks = plan.ntasks;
  a.tasks = (const DagTask*)plan.tasks;
  a.chain_dep = (const uint32_t*)plan.chain;
  a.topo = (const uint32_t*)plan.topo;
  a.flags = h->d_dag_flags.as<uint32_t>();
  a.ctrl = a.flags + nflags;
  a.info = h->d_info.as<int>();
  a.B = 1;
  a.duo = plan.duo ? 1 : 0;
  a.flags_stride = (uint32_t)nflags;
  // development aid: TGP_DAG_TRACE=<file> -- time stamps of every chain phase and task of the LAST update, dumped as
  // uint64 [NB][32] + [ntasks][4] after the stream has drained (tools/dag_trace.py reads it)
  static const char* trace_path = getenv("TGP_DAG_TRACE");
  const size_t trace_words = 32 * (size_t)NB + 4 * (size_t)plan.ntasks;
  if (trace_path) {
    HIPCHK(h, h->d_dag_trace.reserve(trace_words * 8));
    HIPCHK(h, hipMemsetAsync(h->d_dag_trace.p, 0, trace_words * 8, h->stream));
    a.trace = h->d_dag_trace.as<unsigned long long>();
  }
  HIPCHK(h, launch_dag_update(h->stream, a, grid));
  static const char* dump_path = getenv("TGP_DAG_DUMP");  // development aid: the A buffer (partial sums) after the launch
  if (dump_path) {
    std::vector<double> hostA((size_t)Npad * Npad);
    HIPCHK(h, hipMemcpyAsync(hostA.data(), h->d_A.p, hostA.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (FILE* f = fopen(dump_path, "wb")) {
      fwrite(hostA.data(), 8, hostA.size(), f);
      HIPCHK(h, hipMemcpyAsync(hostA.data(), h->d_L.p, hostA.size() * 8, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      fwrite(hostA.data(), 8, hostA.size(), f);
      fclose(f);
    }
  }
  if (trace_path) {
    std::vector<unsigned long long> host(trace_words);
    HIPCHK(h, hipMemcpyAsync(host.data(), h->d_dag_trace.p, trace_words * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (FILE* f = fopen(trace_path, "wb")) {
      const unsigned long long hdr[2] = {(unsigned long long)NB, (unsigned long long)plan.ntasks};
      fwrite(hdr, 8, 2, f);
      fwrite(host.data(), 8, host.size(), f);
      fclose(f);
    }
  }
  return TGP_OK;
}

// Products with few output tiles and a long k (N x P x N, P <= 128: gradients, cross-covariances): split k.
int gemm_tall(tgp_handle h, bool tb, int m, int n, int k, double alpha, const double* A, int64_t lda,
              const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int tri) {
  const int64_t tiles = (int64_t)(m / 64) * (n / 64);
  int nz = 1;
  // up to 16 slices towards 2048 workgroups (rounds 1 - 5: 8 towards 512).  profiles/r06_ksplit.txt: with a slice of each tile's own
  // pruned range the finer split paid from k = 4096 on only (the value-and-gradient call at 80 points 0.252 -> 0.225 ms, neutral at
  // 2048, slower at 1024); with slices of FIXED depth (gemm8_body: empty slices write nothing, ksplit_reduce_kernel sums the live ones)
  // it pays at every size -- 0.132 -> 0.118 ms at N = 2048, 0.104 -> 0.093 at N = 1024 (profiles/r06_ksplit_fixed.txt)
  static const int64_t nz_max_env = getenv("TGP_KSPLIT_MAX") ? atoll(getenv("TGP_KSPLIT_MAX")) : 0;          // (env: development aids)
  static const int64_t nz_target_env = getenv("TGP_KSPLIT_TARGET") ? atoll(getenv("TGP_KSPLIT_TARGET")) : 0;
  const int64_t nz_max = nz_max_env ? nz_max_env : 16, nz_target = nz_target_env ? nz_target_env : 2048;
  if (tiles < 256 && k >= 1024) nz = (int)std::min<int64_t>(nz_max, std::max<int64_t>(1, nz_target / tiles));
  if (nz > 1) {
    HIPCHK(h, h->s_ks.reserve((size_t)nz * m * n * sizeof(double)));
    launch_gemm_ksplit(h->stream, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, nz, h->s_ks.as<double>());
  } else {
    launch_gemm(h->stream, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, false, tri);
  }
  return TGP_OK;
}

constexpr int ACQ_KIND_MAX = TGP_ACQ_GIBBON;  // public kinds: EI, PI, -LCB, AEI, MES, GIBBON
constexpr int ACQ_LOGYVAR = 6;  // internal tail log(var + noise): tgp::ACQ_LOGYVAR of tgp_dev.hpp

// Is the repulsion twin this model + m <= 16 appended rows with equal hyper-parameters?  Then (see
// lowrank_var_kernel) its variance needs m kernel sums per candidate instead of a second sweep.  The answer and
// the weight columns are cached per (this model's data, the twin's data) version pair.
int prepare_repulsion(tgp_handle h) {
  tgp_handle t = h->rep_twin;
  if (h->rep_checked && h->rep_self_version == h->data_version && h->rep_twin_version == t->data_version)
    return TGP_OK;
  h->rep_checked = false;
  h->rep_lowrank = false;
  const int64_t m = t->N - h->N;
  const bool no_lowrank = getenv("TGP_NO_LOWRANK") != nullptr;  // A/B aid (read per check: tests toggle it)
  bool ok = !no_lowrank && m >= 1 && m <= 16 && t->variance == h->variance && t->noise == h->noise && t->ls == h->ls;
  if (ok) {
    HIPCHK(h, h->s_small.reserve(64));
    int* flag = h->s_small.as<int>();
    HIPCHK(h, hipMemsetAsync(flag, 0, sizeof(int), h->stream));
    launch_prefix_differs(h->stream, h->d_X.as<double>(), t->d_X.as<double>(), h->N * h->d, flag);
    int differs = 0;
    HIPCHK(h, hipMemcpyAsync(&differs, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    ok = differs == 0;
  }
  if (ok) {
    HIPCHK(h, h->d_repv.reserve((size_t)t->Npad * m * sizeof(double)));
    launch_rows_to_columns(h->stream, t->d_W.as<double>(), t->Npad, h->N, (int)m, t->Npad, h->d_repv.as<double>());
    h->rep_m = (int)m;
    h->rep_lowrank = true;
  }
  h->rep_self_version = h->data_version;
  h->rep_twin_version = t->data_version;
  h->rep_checked = true;
  return TGP_OK;
}

// Acquisition values of M device-resident candidates into dvals (device, [M]): the fused sweep for the posterior
// tails; sweep -> (mean, var) -> tail kernel for the entropy tails (their S-sample loop stays out of the MFMA
// kernel's register budget; 16 B per candidate of extra HBM traffic against N^2 flops); then the local
// penalization, if one is set.
int acq_values_device(tgp_handle h, int acq_kind, double param, const double* dXq, int64_t M, double* dvals) {
  SweepArgs a{};
  a.m = model_dev(h);
  a.Xq = dXq;
  a.M = M;
  if (acq_kind < TGP_ACQ_MES) {
    a.acq_out = dvals;
    a.acq_kind = acq_kind;
    a.acq_param = param;
    HIPCHK(h, launch_sweep_timed(h, a, false));
  } else {
    if (h->ent_S == 0)
      return fail(h, TGP_ERR_STATE, "entropy-search acquisition needs min-value samples: call tgp_set_min_value_samples");
    HIPCHK(h, h->s_ent.reserve((size_t)3 * M * sizeof(double)));
    double* mean = h->s_ent.as<double>();
    double* var = mean + M;
    double* var_twin = nullptr;
    a.mean_out = mean;
    a.var_out = var;
    a.acq_kind = -1;
    HIPCHK(h, launch_sweep_timed(h, a, false));
    if (acq_kind == TGP_ACQ_GIBBON && h->rep_twin) {
      tgp_handle t = h->rep_twin;
      if (!t->have_data) return fail(h, TGP_ERR_STATE, "the repulsion twin has no data");
      if (int rc = prepare_repulsion(h)) return rc;
      var_twin = var + M;
      if (h->rep_lowrank) {  // rank-m correction of this model's variance
        const int m = h->rep_m;
        HIPCHK(h, h->s_ks.reserve((size_t)M * m * sizeof(double)));
        TrajDev tr{};
        tr.m = model_dev(t);
        tr.F = 0;
        tr.B = m;
        tr.v = h->d_repv.as<double>();
        tr.canonical = 1;
        launch_kernel_sums(h->stream, tr, dXq, M, h->s_ks.as<double>());
        launch_lowrank_var(h->stream, var, h->s_ks.as<double>(), M, m, var_twin);
      } else {  // any other twin: its own variance sweep
        SweepArgs b{};
        b.m = model_dev(t);
        b.Xq = dXq;
        b.M = M;
        b.var_out = var_twin;
        b.acq_kind = -1;
        HIPCHK(h, launch_sweep_timed(h, b, false));
      }
    }
    launch_entropy_tail(h->stream, mean, var, var_twin, M, acq_kind, h->noise, h->d_ent.as<double>(), h->ent_S,
                        h->rep_weight, dvals);
  }
  apply_penalization(h, dvals, dXq, M);
  return TGP_OK;
}

}  // namespace

namespace tgp {
int host_set_device(tgp_handle h) { return set_device(h); }
int host_fail(tgp_handle h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return fail(h, code, "%s", buf);
}
int64_t sweep_grid(const SweepArgs& a, bool joint) {
  if (!joint) return (a.M + SW_BN - 1) / SW_BN;
  const int gp = 64 / a.q;
  return (a.G + 2 * gp - 1) / (2 * gp);
}
}  // namespace tgp

extern "C" {

const char* tgp_version(void) { return "tgp 0.1 (gfx950)"; }

const char* tgp_last_error(tgp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tgp_create(int device_id, int d, int kernel_kind, tgp_handle* out) {
  if (!out) return fail(nullptr, TGP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (d < 1 || d > MAX_D) return fail(nullptr, TGP_ERR_SHAPE, "d must be in 1..%d, got %d", MAX_D, d);
  if (kernel_kind < 0 || kernel_kind > 3)
    return fail(nullptr, TGP_ERR_ARG, "unknown kernel kind %d", kernel_kind);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, TGP_ERR_HIP, "no HIP device available (%s): this engine has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, TGP_ERR_ARG, "device %d out of range (have %d)", device_id, ndev);
  tgp_handle h = new (std::nothrow) tgp_handle_s();
  if (!h) return fail(nullptr, TGP_ERR_ALLOC, "host allocation failed");
  h->device = device_id;
  h->d = d;
  h->dp = dpad_of(d);
  h->kind = kernel_kind;
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipEventCreate(&h->ev0)) != hipSuccess ||
      (e = hipEventCreate(&h->ev1)) != hipSuccess) {
    delete h;
    return fail(nullptr, TGP_ERR_HIP, "device init failed: %s", hipGetErrorString(e));
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
      h->num_cu = prop.multiProcessorCount;
  }
  *out = h;
  return TGP_OK;
}

int tgp_destroy(tgp_handle h) {
  if (!h) return TGP_OK;
  (void)hipSetDevice(h->device);
  // a caller-provided stream (tgp_set_stream) may already be gone when a garbage collector destroys the
  // handle: only the handle's own stream is synchronised / destroyed here; hipFree below synchronises the
  // device before releasing memory in any case
  if (h->own_stream) {
    (void)hipStreamSynchronize(h->own_stream);
    (void)hipStreamDestroy(h->own_stream);
  } else if (h->stream == nullptr) {
    (void)hipStreamSynchronize(nullptr);
  }
  for (DevBuf* b : {&h->d_xn, &h->d_ls, &h->d_X, &h->d_Y, &h->d_Xs, &h->d_A, &h->d_L, &h->d_W, &h->d_alpha,
                    &h->d_err, &h->d_tmp1, &h->d_tmp2, &h->d_info, &h->d_pen, &h->d_ent, &h->d_repv, &h->d_wq, &h->d_rs, &h->d_xsa, &h->s_ent, &h->s_in, &h->s_in2, &h->s_out1,
                    &h->s_out2, &h->s_out3, &h->s_blkv, &h->s_blki, &h->s_small, &h->s_kcache, &h->s_aslab, &h->s_grad, &h->s_ks, &h->s_part, &h->s_rep,
                    &h->s_rep_stats, &h->d_dag_flags, &h->d_dag_trace})
    b->release();

  if (h->rep_host) (void)hipHostFree(h->rep_host);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  (void)hipGetLastError();  // do not leave a cleanup error behind for the thread's next launch check
  return TGP_OK;
}

int tgp_set_stream(tgp_handle h, void* hip_stream) {
  if (!h) return TGP_ERR_ARG;
  h->stream = (hipStream_t)hip_stream;
  return TGP_OK;
}

int tgp_use_private_stream(tgp_handle h) {
  if (!h) return TGP_ERR_ARG;
  if (int rc = set_device(h)) return rc;
  if (!h->own_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  return TGP_OK;
}

int tgp_set_precision(tgp_handle h, int precision) {
  if (!h) return TGP_ERR_ARG;
  if (precision != TGP_PREC_F64 && precision != TGP_PREC_I8X4 && precision != TGP_PREC_I8X5 && precision != TGP_PREC_AUTO)
    return fail(h, TGP_ERR_ARG, "unknown precision %d", precision);
  if (precision == TGP_PREC_I8X5 && h->dp > 16)
    return fail(h, TGP_ERR_ARG, "TGP_PREC_I8X5 supports input dimensions up to 16 (LDS), got %d", h->d);
  h->precision_req = precision;
  h->precision = precision == TGP_PREC_AUTO ? TGP_PREC_F64 : precision;  // AUTO: resolved at the next plain sweep
  h->repair = false;
  auto_restart(h, true);  // AUTO restarts its ladder at four planes
  h->auto_pinned = false;
  return TGP_OK;
}

int tgp_set_auto_sigma(tgp_handle h, double k_sigma) {
  if (!h) return TGP_ERR_ARG;
  if (!(k_sigma > 0.0) || !std::isfinite(k_sigma)) return fail(h, TGP_ERR_ARG, "k_sigma must be positive and finite");
  h->auto_sigma = k_sigma;
  auto_restart(h, true);
  h->auto_pinned = false;
  return TGP_OK;
}

int tgp_get_auto_report(tgp_handle h, int64_t* checked, int64_t* violations, double* worst_ratio, int* demotions, int* level) {
  if (!h) return TGP_ERR_ARG;
  if (int rc = set_device(h)) return rc;
  if (h->precision_req == TGP_PREC_AUTO && h->have_data) {
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the last sweep's report has landed
    HIPCHK(h, resolve_precision(h));
  }
  if (checked) *checked = h->can_checked_total[0] + h->can_checked[0] + h->can_checked_total[1] + h->can_checked[1];
  if (violations) *violations = h->can_viol_total[0] + h->can_viol[0] + h->can_viol_total[1] + h->can_viol[1];
  if (worst_ratio) *worst_ratio = std::max(h->can_worst[0], h->can_worst[1]);
  if (demotions) *demotions = h->can_demotions;
  if (level) *level = h->precision_req == TGP_PREC_AUTO ? h->auto_level : -1;
  return TGP_OK;
}

int tgp_get_auto_strata(tgp_handle h, int64_t* checked2, int64_t* violations2, double* worst_ratio2, int64_t* slack_saved) {
  if (!h) return TGP_ERR_ARG;
  if (int rc = set_device(h)) return rc;
  if (h->precision_req == TGP_PREC_AUTO && h->have_data) {
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the last sweep's report has landed
    HIPCHK(h, resolve_precision(h));
  }
  for (int k = 0; k < 2; ++k) {
    if (checked2) checked2[k] = h->can_checked_total[k] + h->can_checked[k];
    if (violations2) violations2[k] = h->can_viol_total[k] + h->can_viol[k];
    if (worst_ratio2) worst_ratio2[k] = h->can_worst[k];
  }
  if (slack_saved) *slack_saved = h->can_slack_total + h->can_slack;
  return TGP_OK;
}

int tgp_set_variant(tgp_handle h, int variant) {
  if (!h) return TGP_ERR_ARG;
  h->variant = variant;
  return TGP_OK;
}

int tgp_set_update_concurrency(tgp_handle h, int n) {
  if (!h) return TGP_ERR_ARG;
  if (n < 1 || n > 16) return fail(h, TGP_ERR_ARG, "update concurrency must be 1 ... 16, got %d", n);
  h->update_share = n;
  return TGP_OK;
}

static_assert(sizeof(tgp_dag_task) == sizeof(tgp::DagTask), "tgp_dag_task mirrors tgp::DagTask");
int tgp_dag_plan(int nb, int64_t ld, tgp_dag_task* tasks, int64_t cap, int64_t* ntasks, int64_t* n_urgent,
                 uint32_t* chain_dep, uint32_t* order, int flags) {
  if (nb < 1 || nb > 126 || ld < (int64_t)nb * 128 || !ntasks || !n_urgent) return TGP_ERR_ARG;
  const int B = (flags >> 8) & 255;  // > 0: the dispatch list of a batched launch of B members (B ntasks entries)
  if (B > 64) return TGP_ERR_ARG;
  std::vector<tgp::DagTask> t;
  std::vector<uint32_t> c, topo, merged;
  int nu = 0;
  if ((flags & 6) && ((flags & 1) || B > 0)) return TGP_ERR_ARG;   // the split plan and the two-workgroup chain are the single full update's
  tgp::dag_build(nb, ld, t, c, nu, &topo, B > 0 ? std::max(1, (256 - B) / B) : ((flags & 4) ? 254 : 255), (flags & 1) == 0, std::max(1, B),
                 256 - B, &merged, (flags & 2) != 0, (flags & 4) != 0);
  if (B > 0) {
    if (B == 1) tgp::dag_merge_order(topo, 1, merged);
    topo.swap(merged);
  }
  *ntasks = (int64_t)t.size();
  *n_urgent = nu;
  if (cap < (int64_t)t.size() || !tasks || !chain_dep) return TGP_ERR_SHAPE;
  memcpy(tasks, t.data(), t.size() * sizeof(tgp::DagTask));
  memcpy(chain_dep, c.data(), c.size() * sizeof(uint32_t));
  if (order) memcpy(order, topo.data(), topo.size() * sizeof(uint32_t));
  return TGP_OK;
}

/* Is `update` at N training points ONE persistent launch on this handle (size, variant bits, TGP_NO_DAG)?  The host layer
 * sizes its side-by-side evaluations with it instead of restating the rule. */
int tgp_update_is_persistent(tgp_handle h, int64_t N, int* yes) {
  if (!h || !yes || N < 1) return TGP_ERR_ARG;
  const int64_t Npad = ((N + NPAD_MULT - 1) / NPAD_MULT) * NPAD_MULT;
  *yes = dag_applies(h, Npad) ? 1 : 0;
  return TGP_OK;
}

int tgp_get_precision(tgp_handle h, int* requested, int* effective, double* repaired_fraction) {
  if (!h) return TGP_ERR_ARG;
  if (int rc = set_device(h)) return rc;
  if (h->precision_req == TGP_PREC_AUTO) {
    if (!h->have_data) return fail(h, TGP_ERR_STATE, "TGP_PREC_AUTO is resolved per model state: call tgp_set_data first");
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the last sweep's report has landed
    h->auto_pinned = false;
    HIPCHK(h, resolve_precision(h));
  }
  if (requested) *requested = h->precision_req;
  if (effective) *effective = h->precision;
  if (repaired_fraction)
    *repaired_fraction = (h->precision_req == TGP_PREC_AUTO && h->rep_last_M > 0)
                             ? (double)h->rep_last_count / (double)h->rep_last_M : -1.0;
  return TGP_OK;
}

int tgp_set_hyper(tgp_handle h, double variance, const double* lengthscales, double noise_variance,
                  double mean_const) {
  if (!h) return TGP_ERR_ARG;
  if (!lengthscales) return fail(h, TGP_ERR_ARG, "lengthscales is NULL");
  if (!(variance > 0.0) || !(noise_variance > 0.0) || !std::isfinite(mean_const))
    return fail(h, TGP_ERR_ARG, "variance and noise_variance must be positive, mean finite");
  for (int c = 0; c < h->d; ++c)
    if (!(lengthscales[c] > 0.0)) return fail(h, TGP_ERR_ARG, "lengthscale %d must be positive", c);
  if (int rc = set_device(h)) return rc;
  h->variance = variance;
  h->noise = noise_variance;
  h->mean_const = mean_const;
  h->ls.assign(lengthscales, lengthscales + h->d);
  std::vector<double> lsp(h->dp, 1.0);
  for (int c = 0; c < h->d; ++c) lsp[c] = lengthscales[c];
  HIPCHK(h, h->d_ls.reserve(h->dp * sizeof(double)));
  HIPCHK(h, hipMemcpy(h->d_ls.p, lsp.data(), h->dp * sizeof(double), hipMemcpyHostToDevice));
  h->have_hyper = true;
  if (h->precision_req == TGP_PREC_AUTO) {
    // TGP_PREC_AUTO: the epoch ends here (what its canary counted joins the totals, in-flight reports become stale); whether
    // the ladder keeps its rung or restarts is decided at the next sweep from the hyper-parameters in effect THEN
    // (resolve_precision), so that a fit's trial evaluations do not undo a rung the refit would have kept
    h->auto_pinned = false;
    auto_read_report(h);
    auto_fold_counters(h);
    ++h->auto_epoch;
    h->auto_hyper_dirty = true;
  }
  h->have_data = false;  // factorisation is stale
  return TGP_OK;
}

// (Re)build everything that depends on (X, Y, hyper-parameters) from the device copies d_X [N,d], d_Y [N]:
// scaled inputs, K + noise I, L, W = L^-1, Wt, alpha.  `keep_rows` > 0 (a multiple of 64, Npad unchanged):
// rows/columns [0, keep_rows) of L and W are still valid (same inputs, same hyper-parameters) and only
// the tail block is refactorised -- one node step of the recursion with the split at keep_rows:
// L21 = A21 W11^T, A22 -= L21 L21^T, factor A22, W21 = -W22 (L21 W11): O((N - keep) N^2) instead of O(N^3).
// `trial_value` non-null (keep_rows == 0): a TRIAL evaluation of the likelihood at the current hyper-parameters -- where
// the persistent kernel applies only the factor is built (no inverse), z = L^-1 err by block forward substitution and
// *trial_value = 1/2 |z|^2 + sum log L_ii + N/2 log 2 pi; the handle is left WITHOUT a posterior (have_data false).
static int factorise(tgp_handle h, int64_t N, int64_t keep_rows, double* trial_value = nullptr) {
  const int64_t Npad = ((N + NPAD_MULT - 1) / NPAD_MULT) * NPAD_MULT;
  const size_t nn = (size_t)Npad * Npad * sizeof(double);
  const int d = h->d, dp = h->dp;
  HIPCHK(h, h->d_Xs.reserve((size_t)Npad * dp * sizeof(double)));
  HIPCHK(h, h->d_xn.reserve((size_t)Npad * sizeof(double)));
  if (keep_rows == 0) {
    HIPCHK(h, h->d_A.reserve(nn));
    HIPCHK(h, h->d_L.reserve(nn));
    HIPCHK(h, h->d_W.reserve(nn));
  }
  HIPCHK(h, h->d_alpha.reserve(Npad * sizeof(double)));
  HIPCHK(h, h->d_err.reserve(Npad * sizeof(double)));
  HIPCHK(h, h->d_tmp1.reserve(Npad * sizeof(double)));
  HIPCHK(h, h->d_tmp2.reserve(Npad * sizeof(double)));
  HIPCHK(h, h->d_info.reserve(sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->d_info.p, 0, sizeof(int), h->stream));
  h->N = N;
  h->Npad = Npad;
  hipStream_t s = h->stream;
  double* A = h->d_A.as<double>();
  double* L = h->d_L.as<double>();
  double* W = h->d_W.as<double>();
  launch_scale_inputs(s, h->d_X.as<double>(), h->d_ls.as<double>(), h->d_Xs.as<double>(), N, Npad, d, dp);
  launch_row_norms(s, h->d_Xs.as<double>(), h->d_xn.as<double>(), Npad, dp);
  if (keep_rows == 0) {
    launch_assemble_K(s, h->d_Xs.as<double>(), A, N, Npad, dp, h->kind, h->variance, h->noise);
    if (h->zeroed_L != L || h->zeroed_W != W || h->zeroed_npad != Npad) {
      HIPCHK(h, hipMemsetAsync(L, 0, nn, s));
      HIPCHK(h, hipMemsetAsync(W, 0, nn, s));
      h->zeroed_L = L;
      h->zeroed_W = W;
      h->zeroed_npad = Npad;
    }
    static const bool timing = getenv("TGP_TIMING") != nullptr;  // development aid: enqueue vs execution time
    std::chrono::steady_clock::time_point tq0;
    if (timing) { (void)hipStreamSynchronize(s); tq0 = std::chrono::steady_clock::now(); }
    const bool dag = dag_applies(h, Npad);
    if (dag) {
      if (int rc = chol_inv_dag(h, trial_value != nullptr)) return rc;
    } else {
      chol_inv(h, 0, Npad);
    }
    if (timing) {
      const auto tq1 = std::chrono::steady_clock::now();
      (void)hipStreamSynchronize(s);
      const auto tq2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[tgp] chol_inv N=%lld: enqueue %.3f ms, total %.3f ms\n", (long long)Npad,
              std::chrono::duration<double, std::milli>(tq1 - tq0).count(),
              std::chrono::duration<double, std::milli>(tq2 - tq0).count());
    }
  } else {
    const int64_t lo = keep_rows, s2 = Npad - lo;
    // rows [lo, Npad) of K + noise I in a scratch strip addressed with global row indices (d_A holds Wt)
    HIPCHK(h, h->s_grad.reserve((size_t)s2 * Npad * sizeof(double)));
    double* As = h->s_grad.as<double>() - (size_t)lo * Npad;
    launch_assemble_K(s, h->d_Xs.as<double>(), As, N, Npad, dp, h->kind, h->variance, h->noise, lo);
    if (h->zeroed_L != L || h->zeroed_W != W || h->zeroed_npad != Npad) {  // (cannot happen: the append path keeps its buffers)
      HIPCHK(h, hipMemsetAsync(L + (size_t)lo * Npad, 0, (size_t)s2 * Npad * sizeof(double), s));
      HIPCHK(h, hipMemsetAsync(W + (size_t)lo * Npad, 0, (size_t)s2 * Npad * sizeof(double), s));
    }
    double* A21 = As + (size_t)lo * Npad;
    double* L21 = L + (size_t)lo * Npad;
    double* A22 = As + (size_t)lo * Npad + lo;
    // a strip of s2 <= 256 + 63 rows against lo ~ N columns: few output tiles, each walking k = lo alone --
    // the three long-k products split k over up to 8 workgroups per tile (fixed-order reduction)
    if (int rc = gemm_tall(h, true, (int)s2, (int)lo, (int)lo, 1.0, A21, Npad, W, Npad, 0.0, L21, Npad, 1)) return rc;
    if (int rc = gemm_tall(h, true, (int)s2, (int)s2, (int)lo, -1.0, L21, Npad, L21, Npad, 1.0, A22, Npad, 0)) return rc;
    chol_inv(s, FactorWs{As, L, W, Npad, h->d_info.as<int>()}, lo, Npad);
    double* W22 = W + (size_t)lo * Npad + lo;
    double* W21 = W + (size_t)lo * Npad;
    if (int rc = gemm_tall(h, false, (int)s2, (int)lo, (int)lo, 1.0, L21, Npad, W, Npad, 0.0, A21, Npad, 2)) return rc;
    launch_gemm(s, false, (int)s2, (int)lo, (int)s2, -1.0, W22, Npad, A21, Npad, 0.0, W21, Npad, false, 3);
  }
  // err = Y - c (zero padded)  -- gpflow GPRPosterior._precompute: err = Y - mean_function(X)
  launch_center(s, h->d_Y.as<double>(), h->mean_const, h->d_err.as<double>(), N, Npad);
  const bool used_dag = keep_rows == 0 && dag_applies(h, Npad);
  const bool factor_only = trial_value != nullptr && used_dag;
  double* trial_out = nullptr;
  if (factor_only) {
    // z (into the alpha buffer) = L^-1 err; the value kernel's err . alpha is then |z|^2
    uint32_t* const tflags = h->d_tmp1.as<uint32_t>();  // [Npad / 128] words of the Npad doubles
    HIPCHK(h, hipMemsetAsync(tflags, 0, (size_t)(Npad / 128) * sizeof(uint32_t), s));
    launch_block_trsv(s, L, W, Npad, (int)(Npad / 128), h->d_err.as<double>(), h->d_alpha.as<double>(), tflags);
    HIPCHK(h, h->s_small.reserve(64 + (MAX_D + 8) * sizeof(double)));
    trial_out = h->s_small.as<double>() + 8;
    launch_nlml_value(s, model_dev(h), L, h->d_alpha.as<double>(), trial_out);
  } else {
    // Wt (into A, dead now) = masked transpose of W; alpha = Wt (W err)
    launch_transpose_mask(s, W, A, N, Npad);
    launch_trmv(s, W, Npad, Npad, h->d_err.as<double>(), h->d_tmp2.as<double>(), true);
    launch_trmv(s, A, Npad, Npad, h->d_tmp2.as<double>(), h->d_alpha.as<double>(), false);
    // the padding rows of W carry the identity: alpha/tmp there are err_pad = 0 -> stay 0.
    if (trial_value) {  // (below the persistent kernel's sizes a trial is a full update)
      HIPCHK(h, h->s_small.reserve(64 + (MAX_D + 8) * sizeof(double)));
      trial_out = h->s_small.as<double>() + 8;
      launch_nlml_value(s, model_dev(h), L, h->d_err.as<double>(), trial_out);
    }
  }
  int info = 0;
  uint32_t dag_ctrl[4] = {0, 0, 0, 0};
  if (trial_out) HIPCHK(h, hipMemcpyAsync(trial_value, trial_out, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(&info, h->d_info.p, sizeof(int), hipMemcpyDeviceToHost, s));
  if (used_dag)
    HIPCHK(h, hipMemcpyAsync(dag_ctrl, h->d_dag_flags.as<uint32_t>() + (size_t)h->dag_plan[h->dag_last_slot].ntasks +
                                           2 * (size_t)h->dag_plan[h->dag_last_slot].nb,
                             sizeof dag_ctrl, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  if (dag_ctrl[2] != 0)
    return fail(h, TGP_ERR_HIP, "persistent update kernel gave up (code %u) waiting for flag %u of %d tasks + %d chain "
                "steps: a workgroup did not become resident or a dependency is wrong", dag_ctrl[2], dag_ctrl[3],
                h->dag_plan[h->dag_last_slot].ntasks, 2 * h->dag_plan[h->dag_last_slot].nb);
  if (info != 0)
    return fail(h, TGP_ERR_NOT_PD, "Cholesky failed: K + noise*I is not positive definite (pivot %d)",
                info - 1);
  h->have_data = !factor_only;  // (a factor-only trial leaves no posterior behind)
  h->data_version = ++g_data_version;
  return TGP_OK;
}

int tgp_nlml_trial(tgp_handle h, double* value) {
  if (!h || !value) return TGP_ERR_ARG;
  if (!h->have_hyper) return fail(h, TGP_ERR_STATE, "tgp_set_hyper must be called before tgp_nlml_trial");
  if (!h->have_xy) return fail(h, TGP_ERR_STATE, "no data on the device: call tgp_set_data once first");
  if (int rc = set_device(h)) return rc;
  h->have_data = false;
  return factorise(h, h->N, 0, value);
}

// One launch group of a batched trial evaluation, ENQUEUE ONLY: B members through ONE persistent factor-only launch.
// `small` / `infos` / `ctrl_copy` are this group's slices of the call's device block (per member: ls [32], variance,
// noise, mean, value slots -- uploaded by the caller; breakdown reports; the launch's error words): the groups of a call follow each other on the stream without a host round trip (1.2 ms per group when each
// was synchronised: profiles/r04_bo_step.txt) and share the matrices -- stream order keeps them apart.
// The scratch of the batched trial evaluations -- the members' K / L / W (3 N^2 doubles each: 5.8 GiB for fifteen members
// at N = 4096), their vectors and result slots -- is PROCESS-WIDE, one per device, allocated on first use and kept: a BO
// loop fits a fresh or re-attached model every step, and allocating / releasing gigabytes per model cost 10 - 40 ms each
// way (sporadically hundreds).  A call holds the device's mutex from its first launch to its synchronisation.
struct BatchScratch {
  std::mutex mu;
  DevBuf mats, vec, small;
  const void* zeroed = nullptr;
  int64_t zeroed_npad = 0;
  int zeroed_B = 0;
};
static BatchScratch& batch_scratch(int device) {
  static std::mutex table_mu;
  static std::map<int, BatchScratch*> table;  // (never freed: process lifetime)
  std::lock_guard<std::mutex> lk(table_mu);
  BatchScratch*& p = table[device];
  if (!p) p = new BatchScratch();
  return *p;
}

static constexpr size_t TRIAL_SMALL_PER = 40 + (MAX_D + 8);  // ls [32], variance, noise, mean (32 .. 34), value slots from 40
static int nlml_trial_enqueue(tgp_handle h, BatchScratch& bs, int B, double* small, int* infos, uint32_t* ctrl_copy) {
  const int64_t N = h->N, Npad = h->Npad;
  const int d = h->d, dp = h->dp, NB = (int)(Npad / 128);
  const size_t nn = (size_t)Npad * Npad;
  hipStream_t s = h->stream;
  const int grid = h->num_cu;
  const int slot = 2 + (B - 1);
  if (int rc = dag_plan_get(h, slot, NB, Npad, grid, B)) return rc;
  const tgp_handle_s::DagPlan& plan = h->dag_plan[slot];
  h->dag_last_slot = slot;
  const size_t nflags = (size_t)plan.ntasks + 2 * (size_t)NB;
  const size_t state_words = (size_t)B * nflags + DAG_CTRL_WORDS + (size_t)B * plan.ntasks;
  const size_t small_per = TRIAL_SMALL_PER;
  // [B] scaled inputs Xs [Npad][dp]; [B] centred targets err [Npad]; [B] z [Npad]; the trsv flags [B][NB]
  const size_t xs_per = (size_t)Npad * dp;
  double* const mats = bs.mats.as<double>();
  double* const Xs_all = bs.vec.as<double>();
  double* const errs = Xs_all + (size_t)B * xs_per;
  double* const zs = errs + (size_t)B * Npad;
  uint32_t* const tflags = (uint32_t*)(zs + (size_t)B * Npad);
  // every member's scaled inputs / centred targets, then every member's K + noise I: two launches for the group (the
  // per-member scalars -- variance, noise, mean: slots 32 .. 34 of the member's block -- were uploaded with the call)
  launch_batch_prep(s, h->d_X.as<double>(), h->d_Y.as<double>(), small, (int64_t)small_per, B, Xs_all, errs, N, Npad, d, dp);
  launch_assemble_K_batch(s, Xs_all, mats, N, Npad, dp, h->kind, small, (int64_t)small_per, (int64_t)(3 * nn), B);
  static const bool timing = getenv("TGP_TIMING") != nullptr;  // development aid: where a batched launch spends its time
  hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};
  if (timing) {
    for (auto& e : tev) (void)hipEventCreate(&e);
    (void)hipEventRecord(tev[0], s);
  }
  HIPCHK(h, hipMemsetAsync(h->d_dag_flags.p, 0, state_words * sizeof(uint32_t), s));
  DagArgs a{};
  a.Ap = mats;
  a.Lp = mats + nn;
  a.Wp = mats + 2 * nn;
  a.ld = Npad;
  a.NB = NB;
  a.ntasks = plan.ntasks;
  a.tasks = (const DagTask*)plan.tasks;
  a.chain_dep = (const uint32_t*)plan.chain;
  a.topo = (const uint32_t*)plan.topo;
  a.flags = h->d_dag_flags.as<uint32_t>();
  a.ctrl = a.flags + (size_t)B * nflags;
  a.info = infos;
  a.B = B;
  a.mat_stride = (int64_t)(3 * nn);
  a.flags_stride = (uint32_t)nflags;
  if (timing) (void)hipEventRecord(tev[1], s);
  HIPCHK(h, launch_dag_update(s, a, grid));
  if (timing) (void)hipEventRecord(tev[2], s);
  HIPCHK(h, hipMemcpyAsync(ctrl_copy, a.ctrl, 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  // z_b = L_b^-1 err_b: ONE launch for all members (a member is a 32-step chain of block products), then the values
  HIPCHK(h, hipMemsetAsync(tflags, 0, (size_t)B * NB * sizeof(uint32_t), s));
  launch_block_trsv(s, mats + nn, mats + 2 * nn, Npad, NB, errs, zs, tflags, B, (int64_t)(3 * nn));
  {
    ModelDev m{};
    m.kind = h->kind;
    m.d = d;
    m.dp = dp;
    m.N = N;
    m.Npad = Npad;
    launch_nlml_value_batch(s, m, mats + nn, zs, small + 40, B, (int64_t)(3 * nn), Npad, (int64_t)small_per);
  }
  if (timing) {
    (void)hipEventRecord(tev[3], s);
    (void)hipEventSynchronize(tev[3]);
    float t01 = 0, t12 = 0, t23 = 0;
    (void)hipEventElapsedTime(&t01, tev[0], tev[1]);
    (void)hipEventElapsedTime(&t12, tev[1], tev[2]);
    (void)hipEventElapsedTime(&t23, tev[2], tev[3]);
    fprintf(stderr, "[tgp] trial batch B=%d N=%lld: flags memset %.3f ms, persistent launch %.3f ms, trsv + values %.3f ms\n", B,
            (long long)Npad, t01, t12, t23);
    for (auto& e : tev) (void)hipEventDestroy(e);
  }
  return TGP_OK;
}

int tgp_nlml_trial_batch(tgp_handle h, const double* hypers, int B, double* values, int* status) {
  if (!h || !hypers || !values || !status) return TGP_ERR_ARG;
  if (B < 1) return fail(h, TGP_ERR_SHAPE, "B must be >= 1, got %d", B);
  if (!h->have_xy) return fail(h, TGP_ERR_STATE, "no data on the device: call tgp_set_data once first");
  if (int rc = set_device(h)) return rc;
  const int d = h->d;
  for (int b = 0; b < B; ++b) {
    const double* hb = hypers + (size_t)b * (d + 3);
    bool ok = hb[0] > 0.0 && hb[1 + d] > 0.0 && std::isfinite(hb[2 + d]);
    for (int c = 0; c < d; ++c) ok = ok && hb[1 + c] > 0.0;
    if (!ok) return fail(h, TGP_ERR_ARG, "member %d: variance, lengthscales and noise must be positive, the mean finite", b);
  }
  const int64_t Npad = ((h->N + NPAD_MULT - 1) / NPAD_MULT) * NPAD_MULT;
  // the members one after the other on the handle itself (tgp_nlml_trial's arithmetic), then its hyper-parameters restored:
  // below the persistent kernel's sizes, where a factorisation is a chain of small launches -- and the way out when the
  // batched launch cannot be had (no memory for even one member's matrices, too few compute units for chains AND workers)
  auto one_by_one = [&]() -> int {
    const double v0 = h->variance, n0 = h->noise, c0 = h->mean_const;
    const std::vector<double> ls0 = h->ls;
    const bool had = h->have_data;
    int rc_all = TGP_OK;
    for (int b = 0; b < B && rc_all == TGP_OK; ++b) {
      const double* hb = hypers + (size_t)b * (d + 3);
      if (int rc = tgp_set_hyper(h, hb[0], hb + 1, hb[1 + d], hb[2 + d])) return rc;
      const int rc = tgp_nlml_trial(h, values + b);
      status[b] = rc == TGP_ERR_NOT_PD ? TGP_ERR_NOT_PD : TGP_OK;
      if (rc == TGP_ERR_NOT_PD) values[b] = __builtin_nan("");
      else if (rc != TGP_OK) rc_all = rc;
    }
    if (int rc = tgp_set_hyper(h, v0, ls0.data(), n0, c0)) return rc;
    if (rc_all != TGP_OK) return rc_all;
    if (had) return factorise(h, h->N, 0);  // the handle's own posterior, as it was
    return TGP_OK;
  };
  if (!dag_applies(h, Npad) || Npad != h->Npad) return one_by_one();
  // members per launch: up to 16, at most a quarter of the compute units (every member's chain is a workgroup of its own
  // and the launch needs workers beside them: tgp_kernels_dag.hip), at most 12 GiB of matrices and at most what the device
  // has free; the members spread evenly over the launches (90 draws = six launches of fifteen: a launch has fixed costs
  // -- the ramp-up and the chain-bound tail of the persistent kernel, the block forward substitution -- of ~1.2 ms at
  // N = 4096 whatever its size)
  const int64_t Np = h->Npad;
  const int dp = h->dp, NB = (int)(Np / 128);
  const size_t nn = (size_t)Np * Np, per = 3 * nn * sizeof(double);
  const size_t small_per = TRIAL_SMALL_PER;
  BatchScratch& bs = batch_scratch(h->device);
  std::unique_lock<std::mutex> scratch_lock(bs.mu);
  // whatever leaves this function early from here on first waits for what it has enqueued: the scratch is shared
  struct DrainOnExit {
    hipStream_t s;
    bool armed = false;
    ~DrainOnExit() { if (armed) (void)hipStreamSynchronize(s); }
  } drain{h->stream};
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
  (void)hipGetLastError();
  const size_t budget = std::min<size_t>((size_t)12 << 30, bs.mats.cap + free_b - free_b / 5);  // (a fifth of what is free stays free)
  // Members per launch group.  From N = 2048 on a group of 15 is bound by tile-task throughput and a larger one buys nothing but
  // scratch (measured, profiles/r06_trial_groups.txt: 12.5 ms for 90 draws at N = 2048 either way); up to N = 1024 a member's launch
  // is its CHAIN (<= 8 steps of ~54 us) and its tile tasks fill a fraction of the workgroups: 90 draws as two launches of 45
  // instead of six of 15 -- find_best_model_initialization(90) 3.55 -> 2.55 ms at N = 512, 5.6 -> 4.2 ms at N = 1024
  constexpr int TRIAL_BATCH_MAX = 48;
  static_assert(sizeof(((tgp_handle_s*)nullptr)->dag_plan) / sizeof(tgp_handle_s::DagPlan) == 2 + TRIAL_BATCH_MAX, "plan slots");
  static const int bmax_env = getenv("TGP_TRIAL_BATCH_MAX") ? atoi(getenv("TGP_TRIAL_BATCH_MAX")) : 0;   // (development aid)
  const bool chain_bound = NB <= 8;
  const int bcap_size = bmax_env > 0 ? std::min(bmax_env, TRIAL_BATCH_MAX) : (chain_bound ? 45 : 16);
  const size_t budget_size = chain_bound ? std::min<size_t>(budget, (size_t)6 << 30) : budget;
  int blimit = (int)std::min<size_t>((size_t)bcap_size, budget_size / per);
  blimit = std::min(blimit, std::max(1, h->num_cu / 4));
  if (blimit < 1 || h->num_cu < 2 * blimit + 2) {
    scratch_lock.unlock();
    return one_by_one();
  }
  int groups = 0, bmax = 0, bcap = 0;
  for (;;) {  // the scratch of ONE launch group; a failed allocation halves the group
    groups = (B + blimit - 1) / blimit;
    bmax = (B + groups - 1) / groups;
    bcap = std::min(bmax, B);
    hipError_t ea = bs.mats.reserve((size_t)bcap * per);
    if (ea == hipSuccess)
      ea = bs.vec.reserve(((size_t)bcap * ((size_t)Np * dp + 2 * (size_t)Np) + (size_t)bcap * NB) * sizeof(double));
    if (ea == hipSuccess) break;
    (void)hipGetLastError();
    bs.zeroed = nullptr;  // (reserve released the old buffer before it failed)
    if (ea != hipErrorOutOfMemory) return fail(h, TGP_ERR_HIP, "batch scratch: %s", hipGetErrorString(ea));
    if (blimit == 1) {
      scratch_lock.unlock();
      return one_by_one();
    }
    blimit = (blimit + 1) / 2;
  }
  // the per-member result block of the WHOLE call: [B] (ls, value slots), [B] breakdown reports, [groups] error words
  size_t max_state = 0;
  for (int bb = 1; bb <= bcap; ++bb) {  // (plans are built lazily: size the state for the largest group by its own plan)
    if (bb != bcap && bb != (B % bmax == 0 ? bcap : B % bmax)) continue;
    if (int rc = dag_plan_get(h, 2 + (bb - 1), NB, Np, h->num_cu, bb)) return rc;
    const size_t nt = (size_t)h->dag_plan[2 + (bb - 1)].ntasks;
    max_state = std::max(max_state, (size_t)bb * (nt + 2 * (size_t)NB) + DAG_CTRL_WORDS + (size_t)bb * nt);
  }
  HIPCHK(h, h->d_dag_flags.reserve(max_state * sizeof(uint32_t)));
  const size_t small_doubles = (size_t)B * small_per + (size_t)B + (size_t)2 * groups + 8;
  HIPCHK(h, bs.small.reserve(small_doubles * sizeof(double)));
  double* const mats = bs.mats.as<double>();
  double* const small = bs.small.as<double>();
  int* const infos = (int*)(small + (size_t)B * small_per);
  uint32_t* const ctrls = (uint32_t*)(small + (size_t)B * small_per + B);
  drain.armed = true;
  if (bs.zeroed != mats || bs.zeroed_npad != Np || bs.zeroed_B < bcap) {
    // the factorisation only ever writes zeros above the diagonals: wiped once per allocation and size (as d_L / d_W)
    HIPCHK(h, hipMemsetAsync(mats, 0, (size_t)bcap * per, h->stream));
    bs.zeroed = mats;
    bs.zeroed_npad = Np;
    bs.zeroed_B = bcap;
  }
  std::vector<double> hsmall(small_doubles, 0.0);  // (also zeroes the breakdown reports and the error words)
  for (int b = 0; b < B; ++b) {
    const double* hb = hypers + (size_t)b * (d + 3);
    for (int c = 0; c < dp; ++c) hsmall[(size_t)b * small_per + c] = c < d ? hb[1 + c] : 1.0;
    hsmall[(size_t)b * small_per + 32] = hb[0];      // variance
    hsmall[(size_t)b * small_per + 33] = hb[1 + d];  // noise variance
    hsmall[(size_t)b * small_per + 34] = hb[2 + d];  // constant mean
  }
  HIPCHK(h, hipMemcpyAsync(small, hsmall.data(), hsmall.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  for (int g = 0; g < groups; ++g) {
    const int b0 = g * bmax, nb = std::min(bmax, B - b0);
    if (int rc = nlml_trial_enqueue(h, bs, nb, small + (size_t)b0 * small_per, infos + b0, ctrls + 4 * g)) return rc;
  }
  std::vector<double> hout(small_doubles);
  HIPCHK(h, hipMemcpyAsync(hout.data(), small, hout.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  drain.armed = false;
  HIPCHK(h, hipGetLastError());
  const int* hinfo = (const int*)(hout.data() + (size_t)B * small_per);
  const uint32_t* hctrl = (const uint32_t*)(hout.data() + (size_t)B * small_per + B);
  for (int g = 0; g < groups; ++g)
    if (hctrl[4 * g + 2] != 0)
      return fail(h, TGP_ERR_HIP, "batched persistent factorisation gave up (code %u) waiting for flag %u (launch group %d of %d)",
                  hctrl[4 * g + 2], hctrl[4 * g + 3], g, groups);
  for (int b = 0; b < B; ++b) {
    status[b] = hinfo[b] != 0 ? TGP_ERR_NOT_PD : TGP_OK;
    values[b] = hinfo[b] != 0 ? __builtin_nan("") : hout[(size_t)b * small_per + 40];
  }
  return TGP_OK;
}

int tgp_release_scratch(int device_id) {
  // the process-wide scratch of the batched trial evaluations on this device (gigabytes, kept between fits on purpose):
  // waits for a call in progress, frees it; the next batched fit allocates again
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
    (void)hipGetLastError();
    return TGP_ERR_ARG;
  }
  BatchScratch& bs = batch_scratch(device_id);
  std::lock_guard<std::mutex> lk(bs.mu);
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(device_id);
  bs.mats.release();
  bs.vec.release();
  bs.small.release();
  bs.zeroed = nullptr;
  bs.zeroed_npad = 0;
  bs.zeroed_B = 0;
  (void)hipSetDevice(prev);
  (void)hipGetLastError();
  return TGP_OK;
}

int tgp_set_data(tgp_handle h, const double* X, const double* Y, int64_t N, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_hyper) return fail(h, TGP_ERR_STATE, "tgp_set_hyper must be called before tgp_set_data");
  if (!X || !Y) return fail(h, TGP_ERR_ARG, "X / Y is NULL");
  if (N < 1) return fail(h, TGP_ERR_SHAPE, "N must be >= 1, got %lld", (long long)N);
  if (int rc = set_device(h)) return rc;
  h->have_data = false;
  const int d = h->d;
  HIPCHK(h, h->d_X.reserve((size_t)N * d * sizeof(double)));
  HIPCHK(h, h->d_Y.reserve((size_t)N * sizeof(double)));
  const hipMemcpyKind kindcp = where == TGP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(h, hipMemcpyAsync(h->d_X.p, X, (size_t)N * d * sizeof(double), kindcp, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_Y.p, Y, (size_t)N * sizeof(double), kindcp, h->stream));
  h->have_xy = true;
  return factorise(h, N, 0);
}

int tgp_append_data(tgp_handle h, const double* Xnew, const double* Ynew, int64_t k, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "no factorisation to extend: call tgp_set_data first");
  if (!Xnew || !Ynew) return fail(h, TGP_ERR_ARG, "Xnew / Ynew is NULL");
  if (k < 1) return fail(h, TGP_ERR_SHAPE, "k must be >= 1, got %lld", (long long)k);
  if (int rc = set_device(h)) return rc;
  const int d = h->d;
  const int64_t N0 = h->N, N = N0 + k, Npad0 = h->Npad;
  h->have_data = false;
  HIPCHK(h, h->d_X.grow_keep((size_t)N * d * sizeof(double), (size_t)N0 * d * sizeof(double), h->stream));
  HIPCHK(h, h->d_Y.grow_keep((size_t)N * sizeof(double), (size_t)N0 * sizeof(double), h->stream));
  const hipMemcpyKind kindcp = where == TGP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(h, hipMemcpyAsync(h->d_X.as<double>() + (size_t)N0 * d, Xnew, (size_t)k * d * sizeof(double), kindcp, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_Y.as<double>() + N0, Ynew, (size_t)k * sizeof(double), kindcp, h->stream));
  const int64_t Npad = ((N + NPAD_MULT - 1) / NPAD_MULT) * NPAD_MULT;
  // the factor of the first floor(N0 / 64) * 64 rows survives when the padded size does not change
  const int64_t keep = (Npad == Npad0) ? (N0 / LEAF) * LEAF : 0;
  return factorise(h, N, keep);
}

int tgp_clone_from(tgp_handle dst, tgp_handle src) {
  if (!dst || !src) return TGP_ERR_ARG;
  if (dst == src) return TGP_OK;
  if (dst->d != src->d || dst->kind != src->kind)
    return fail(dst, TGP_ERR_SHAPE, "clone needs equal input dimension and kernel (dst d=%d kind=%d, src d=%d kind=%d)",
                dst->d, dst->kind, src->d, src->kind);
  if (dst->device != src->device) return fail(dst, TGP_ERR_ARG, "clone across devices is not supported");
  if (!src->have_hyper) return fail(dst, TGP_ERR_STATE, "source has no hyper-parameters");
  if (int rc = set_device(dst)) return rc;
  dst->have_data = false;
  dst->variance = src->variance;
  dst->noise = src->noise;
  dst->mean_const = src->mean_const;
  dst->ls = src->ls;
  dst->have_hyper = true;
  if (dst->precision_req == TGP_PREC_AUTO) {
    // the hyper-parameters changed under the ladder: the epoch ends as in tgp_set_hyper; a source on the same
    // arithmetic hands its rung over (a fantasised copy of a model that left four planes starts where the model is),
    // otherwise the next sweep decides from the copied hyper-parameters
    dst->auto_pinned = false;
    auto_read_report(dst);
    auto_fold_counters(dst);
    ++dst->auto_epoch;
    if (src->precision_req == TGP_PREC_AUTO) {
      dst->auto_level = src->auto_level;
      dst->auto_hyp = src->auto_hyp;
    }
    dst->auto_hyper_dirty = true;
  }
  hipStream_t s = dst->stream;
  auto copy = [&](DevBuf& to, const DevBuf& from, size_t bytes) -> hipError_t {
    if (bytes == 0) return hipSuccess;
    hipError_t e = to.reserve(bytes);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(to.p, from.p, bytes, hipMemcpyDeviceToDevice, s);
  };
  HIPCHK(dst, copy(dst->d_ls, src->d_ls, (size_t)src->d * sizeof(double)));
  if (src->have_data) {
    // every entry point of the source synchronises its stream before returning: its state is complete
    const size_t N = (size_t)src->N, Npad = (size_t)src->Npad, nn = Npad * Npad * sizeof(double);
    HIPCHK(dst, copy(dst->d_X, src->d_X, N * src->d * sizeof(double)));
    HIPCHK(dst, copy(dst->d_Y, src->d_Y, N * sizeof(double)));
    HIPCHK(dst, copy(dst->d_Xs, src->d_Xs, Npad * src->dp * sizeof(double)));
    HIPCHK(dst, copy(dst->d_xn, src->d_xn, Npad * sizeof(double)));
    HIPCHK(dst, copy(dst->d_A, src->d_A, nn));
    HIPCHK(dst, copy(dst->d_L, src->d_L, nn));
    HIPCHK(dst, copy(dst->d_W, src->d_W, nn));
    HIPCHK(dst, copy(dst->d_alpha, src->d_alpha, Npad * sizeof(double)));
    HIPCHK(dst, copy(dst->d_err, src->d_err, Npad * sizeof(double)));
    dst->N = src->N;
    dst->Npad = src->Npad;
    dst->zeroed_L = src->zeroed_L == src->d_L.p ? dst->d_L.p : nullptr;  // the copies carry the zeros along
    dst->zeroed_W = src->zeroed_W == src->d_W.p ? dst->d_W.p : nullptr;
    dst->zeroed_npad = src->zeroed_npad;
    HIPCHK(dst, hipStreamSynchronize(s));
    dst->have_data = true;
    dst->have_xy = true;  // X, Y came along: trial evaluations on the clone are legitimate
    dst->data_version = ++g_data_version;
  }
  return TGP_OK;
}

int tgp_set_penalization(tgp_handle h, int kind, const double* pending, const double* radius, const double* scale,
                         int64_t P) {
  if (!h) return TGP_ERR_ARG;
  if (kind < 0 || kind > 2) return fail(h, TGP_ERR_ARG, "unknown penalizer kind %d", kind);
  if (kind == 0 || P == 0) {
    h->pen_kind = 0;
    h->pen_P = 0;
    return TGP_OK;
  }
  if (P < 0 || P > 1024) return fail(h, TGP_ERR_SHAPE, "number of pending points must be in 0..1024, got %lld", (long long)P);
  if (!pending || !radius || !scale) return fail(h, TGP_ERR_ARG, "pending / radius / scale is NULL");
  if (int rc = set_device(h)) return rc;
  h->pen_kind = 0;
  const size_t np = (size_t)P * h->d;
  HIPCHK(h, h->d_pen.reserve((np + 2 * (size_t)P) * sizeof(double)));
  double* dp = h->d_pen.as<double>();
  HIPCHK(h, hipMemcpyAsync(dp, pending, np * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dp + np, radius, (size_t)P * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(dp + np + P, scale, (size_t)P * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // the host arrays may be released on return
  h->pen_kind = kind;
  h->pen_P = (int)P;
  return TGP_OK;
}

int tgp_set_min_value_samples(tgp_handle h, const double* samples, int S) {
  if (!h) return TGP_ERR_ARG;
  if (S < 0 || S > 4096) return fail(h, TGP_ERR_SHAPE, "number of min-value samples must be in 0..4096, got %d", S);
  if (S == 0) {
    h->ent_S = 0;
    return TGP_OK;
  }
  if (!samples) return fail(h, TGP_ERR_ARG, "samples is NULL");
  if (int rc = set_device(h)) return rc;
  h->ent_S = 0;
  HIPCHK(h, h->d_ent.reserve((size_t)S * sizeof(double)));
  HIPCHK(h, hipMemcpyAsync(h->d_ent.p, samples, (size_t)S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->ent_S = S;
  return TGP_OK;
}

int tgp_set_repulsion(tgp_handle h, tgp_handle twin, double weight) {
  if (!h) return TGP_ERR_ARG;
  if (!twin) {
    h->rep_twin = nullptr;
    h->rep_weight = 0.0;
    return TGP_OK;
  }
  if (twin == h) return fail(h, TGP_ERR_ARG, "the repulsion twin must be a different handle");
  if (twin->d != h->d || twin->kind != h->kind || twin->device != h->device)
    return fail(h, TGP_ERR_SHAPE, "the repulsion twin must share input dimension, kernel and device");
  if (!twin->have_data) return fail(h, TGP_ERR_STATE, "the repulsion twin has no data");
  if (!(weight >= 0.0)) return fail(h, TGP_ERR_ARG, "repulsion weight must be >= 0");
  h->rep_twin = twin;
  h->rep_weight = weight;
  return TGP_OK;
}

int tgp_penalization_values(tgp_handle h, const double* Xq, int64_t M, double* out, int where) {
  if (!h) return TGP_ERR_ARG;
  if (h->pen_kind == 0 || h->pen_P == 0) return fail(h, TGP_ERR_STATE, "no penalization set: call tgp_set_penalization first");
  if (M < 0 || (M > 0 && (!Xq || !out))) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (M == 0) return TGP_OK;
  if (int rc = set_device(h)) return rc;
  const double* dXq;
  double* dout;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, out, M, where, &dout)) return rc;
  const double* pend = h->d_pen.as<double>();
  launch_penalize(h->stream, dout, dXq, M, h->d, h->pen_kind, h->pen_P, pend, pend + (size_t)h->pen_P * h->d,
                  pend + (size_t)h->pen_P * (h->d + 1), true);
  if (int rc = stage_out_finish(h, dout, out, M, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_nlml(tgp_handle h, double* value, double* grad) {
  if (!h || !value) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "no factorisation: call tgp_set_data first");
  if (int rc = set_device(h)) return rc;
  const int64_t Npad = h->Npad;
  const int np = h->d + 4;
  HIPCHK(h, h->s_small.reserve(64 + (MAX_D + 8) * sizeof(double)));
  double* out = h->s_small.as<double>() + 8;
  if (grad) {
    HIPCHK(h, h->s_blkv.reserve((size_t)nlml_blocks(Npad) * (MAX_D + 2) * sizeof(double)));
    HIPCHK(h, h->s_grad.reserve((size_t)Npad * Npad * sizeof(double)));
    double* Kinv = h->s_grad.as<double>();
    // Kinv = W^T W  (Wt is in d_A, W in d_W; both carry explicit zeros outside their triangles)
    // lower tiles only (the reduction uses the symmetry); within the lower triangle the k range
    // [tm T, N) shrinks with the tile row, so the row-by-row order of the live-tile grid is longest-first
    launch_gemm(h->stream, false, (int)Npad, (int)Npad, (int)Npad, 1.0, h->d_A.as<double>(), Npad, h->d_W.as<double>(),
                Npad, 0.0, Kinv, Npad, true, 4, Npad <= 4096 ? 1 : 0);
    launch_nlml(h->stream, model_dev(h), Kinv, h->d_L.as<double>(), h->d_err.as<double>(), h->s_blkv.as<double>(), out);
  } else {  // value only: 1/2 err^T alpha + sum log L_ii + N/2 log(2 pi)
    launch_nlml_value(h->stream, model_dev(h), h->d_L.as<double>(), h->d_err.as<double>(), out);
  }
  std::vector<double> host((size_t)np);
  HIPCHK(h, hipMemcpyAsync(host.data(), out, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  *value = host[0];
  if (grad)
    for (int c = 0; c < h->d + 3; ++c) grad[c] = host[1 + c];
  return TGP_OK;
}

int tgp_get_sizes(tgp_handle h, int64_t* N, int* d) {
  if (!h) return TGP_ERR_ARG;
  if (N) *N = h->have_data ? h->N : 0;
  if (d) *d = h->d;
  return TGP_OK;
}

int tgp_get_factor(tgp_handle h, double* L, double* Winv, double* alpha, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "no factorisation: call tgp_set_data first");
  if (int rc = set_device(h)) return rc;
  const hipMemcpyKind k = where == TGP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const size_t rowb = (size_t)h->N * sizeof(double);
  if (L) HIPCHK(h, hipMemcpy2DAsync(L, rowb, h->d_L.p, (size_t)h->Npad * sizeof(double), rowb, h->N, k, h->stream));
  if (Winv) HIPCHK(h, hipMemcpy2DAsync(Winv, rowb, h->d_W.p, (size_t)h->Npad * sizeof(double), rowb, h->N, k, h->stream));
  if (alpha) HIPCHK(h, hipMemcpyAsync(alpha, h->d_alpha.p, rowb, k, h->stream));
  return sync(h);
}

// predict (mean, variance) at a handful of points: K*^T, ONE skinny triangular product W K* and a two-pass tail -- the arrays and
// kernels of tgp_acq_value_grad -- instead of a sweep launch whose workgroup walks all of W for its 64 candidates (round 6:
// 1.64 -> 0.15 ms for 128 points at N = 4096, 1.68 -> 0.93 ms for 2048; the greedy batch builders predict at their pending points once per element).
// Exact float64 whatever tgp_set_precision says.  tgp_set_variant bit 10 keeps the sweep (A/B, tests).
// (measured, profiles/r06_predict_small_threshold.txt: the product wins up to ~4096 points at every N -- 0.15 / 0.42 / 0.93 ms at
// 128 / 1024 / 2048 points against the sweep launch's 1.64 / 1.65 / 1.68 ms at N = 4096, 0.34 against 5.12 ms for 128 points at
// N = 8192 -- and costs 2 Npad Mpad doubles of scratch: 2048 points it is)
static const int64_t SMALL_PREDICT_M = getenv("TGP_SMALL_PREDICT_M") ? atoll(getenv("TGP_SMALL_PREDICT_M")) : 2048;   // (env: development aid)
static int predict_small(tgp_handle h, const double* dXq, int64_t M, double* dmean, double* dvar) {
  const int64_t Ppad = ((M + 63) / 64) * 64, Npad = h->Npad;
  HIPCHK(h, h->s_grad.reserve(((size_t)2 * Npad * Ppad + predict_small_scratch_doubles(Ppad)) * sizeof(double)));
  double* B = h->s_grad.as<double>();
  double* C1 = B + (size_t)Npad * Ppad;
  double* part = C1 + (size_t)Npad * Ppad;
  const ModelDev m = model_dev(h);
  launch_kstar_t(h->stream, m, dXq, M, Ppad, B);
  if (dvar)
    if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B, Ppad, 0.0, C1, Ppad, 3))
      return rc;
  launch_predict_small_tail(h->stream, m, M, Ppad, B, C1, part, dmean, dvar);
  h->last_launches = 0;
  h->last_ms = -1.0;
  return TGP_OK;
}

static int sweep_common(tgp_handle h, const double* Xq, int64_t M, double* mean, double* var, double* acq,
                        int acq_kind, double param, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (M < 0) return fail(h, TGP_ERR_SHAPE, "M must be >= 0");
  if (M > 0 && !Xq) return fail(h, TGP_ERR_ARG, "Xq is NULL");
  if (int rc = set_device(h)) return rc;
  h->last_launches = 0;
  h->last_ms = 0.0;
  if (M == 0) return TGP_OK;
  const double* dXq;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
  SweepArgs a{};
  a.m = model_dev(h);
  a.Xq = dXq;
  a.M = M;
  if (int rc = stage_out_prepare(h, h->s_out1, mean, M, where, &a.mean_out)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out2, var, M, where, &a.var_out)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out3, acq, M, where, &a.acq_out)) return rc;
  a.acq_kind = acq_kind;
  a.acq_param = param;
  // (not when a sweep policy was asked for by hand -- variant bits 0, 1, 3 -- or another arithmetic than float64 was set: those
  // calls mean the sweep)
  if (acq_kind < 0 && !acq && M <= SMALL_PREDICT_M && h->precision_req == TGP_PREC_F64 &&
      !(h->variant & (VARIANT_SWEEP_SMALL_PREDICT | VARIANT_NO_SPLIT | VARIANT_FORCE_SPLIT | VARIANT_REG_STAGING))) {
    if (int rc = predict_small(h, dXq, M, a.mean_out, a.var_out)) return rc;
    if (int rc = stage_out_finish(h, a.mean_out, mean, M, where)) return rc;
    if (int rc = stage_out_finish(h, a.var_out, var, M, where)) return rc;
    if (int rc = sync(h)) return rc;
    HIPCHK(h, hipGetLastError());
    return TGP_OK;
  }
  for (int attempt = 0;; ++attempt) {
    HIPCHK(h, launch_sweep_timed(h, a, false));
    if (acq_kind >= 0) apply_penalization(h, a.acq_out, dXq, M);
    if (int rc = stage_out_finish(h, a.mean_out, mean, M, where)) return rc;
    if (int rc = stage_out_finish(h, a.var_out, var, M, where)) return rc;
    if (int rc = stage_out_finish(h, a.acq_out, acq, M, where)) return rc;
    if (int rc = sync(h)) return rc;
    HIPCHK(h, hipGetLastError());
    if (attempt < 2 && auto_canary_tripped(h)) continue;  // TGP_PREC_AUTO: a sample broke its bound -> next rung, again
    break;
  }
  return TGP_OK;
}

int tgp_predict(tgp_handle h, const double* Xq, int64_t M, double* mean, double* var, int where) {
  return sweep_common(h, Xq, M, mean, var, nullptr, -1, 0.0, where);
}

int tgp_predict_mean(tgp_handle h, const double* Xq, int64_t M, double* mean, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (M < 0 || (M > 0 && (!Xq || !mean))) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (M == 0) return TGP_OK;
  if (int rc = set_device(h)) return rc;
  const double* dXq;
  double* dmean;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, mean, M, where, &dmean)) return rc;
  launch_predict_mean(h->stream, model_dev(h), dXq, M, dmean);
  if (int rc = stage_out_finish(h, dmean, mean, M, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_eta(tgp_handle h, double* eta) {
  if (!h || !eta) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (int rc = set_device(h)) return rc;
  HIPCHK(h, h->s_out1.reserve(h->N * sizeof(double)));
  HIPCHK(h, h->s_small.reserve(64));
  launch_predict_mean(h->stream, model_dev(h), h->d_X.as<double>(), h->N, h->s_out1.as<double>());
  launch_min_value(h->stream, h->s_out1.as<double>(), h->N, h->s_small.as<double>());
  HIPCHK(h, hipMemcpyAsync(eta, h->s_small.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_acq_values(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t M, double* out,
                   int where) {
  if (acq_kind < 0 || acq_kind > ACQ_KIND_MAX) return fail(h, TGP_ERR_ARG, "unknown acquisition kind %d", acq_kind);
  if (M > 0 && !out) return fail(h, TGP_ERR_ARG, "out is NULL");
  if (h && acq_kind >= TGP_ACQ_MES) {  // entropy tails: sweep + tail kernel
    if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
    if (M < 0) return fail(h, TGP_ERR_SHAPE, "M must be >= 0");
    if (M == 0) return TGP_OK;
    if (!Xq) return fail(h, TGP_ERR_ARG, "Xq is NULL");
    if (int rc = set_device(h)) return rc;
    const double* dXq;
    double* dout;
    if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
    if (int rc = stage_out_prepare(h, h->s_out3, out, M, where, &dout)) return rc;
    for (int attempt = 0;; ++attempt) {
      if (int rc = acq_values_device(h, acq_kind, param, dXq, M, dout)) return rc;
      if (int rc = stage_out_finish(h, dout, out, M, where)) return rc;
      if (int rc = sync(h)) return rc;
      HIPCHK(h, hipGetLastError());
      if (attempt < 2 && auto_canary_tripped(h)) continue;
      break;
    }
    return TGP_OK;
  }
  return sweep_common(h, Xq, M, nullptr, nullptr, out, acq_kind, param, where);
}

int tgp_acq_value_grad(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t P, double* val,
                       double* grad, int where) {
  if (!h) return TGP_ERR_ARG;
  if (acq_kind < 0 || acq_kind > ACQ_KIND_MAX) return fail(h, TGP_ERR_ARG, "unknown acquisition kind %d", acq_kind);
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (P < 0 || (P > 0 && (!Xq || !val || !grad))) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (P == 0) return TGP_OK;
  if (int rc = set_device(h)) return rc;
  const int64_t Ppad = ((P + 63) / 64) * 64, Npad = h->Npad;
  const double* dXq;
  double *dval, *dgrad;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)P * h->d, where, &dXq)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, val, P, where, &dval)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out2, grad, (size_t)P * h->d, where, &dgrad)) return rc;
  const int64_t Nscratch = (acq_kind == TGP_ACQ_GIBBON && h->rep_twin) ? std::max(Npad, h->rep_twin->Npad) : Npad;
  HIPCHK(h, h->s_grad.reserve(((size_t)3 * Nscratch * Ppad + grad_tail_scratch_doubles(Ppad)) * sizeof(double)));
  double* B = h->s_grad.as<double>();
  double* C1 = B + (size_t)Npad * Ppad;
  double* Z = C1 + (size_t)Npad * Ppad;
  double* const gpart = B + (size_t)3 * Nscratch * Ppad;   // the gradient tail's partial sums
  const ModelDev m = model_dev(h);
  launch_kstar_t(h->stream, m, dXq, P, Ppad, B);
  // C1 = W B (W = L^-1, lower triangular incl. explicit zeros), Z = W^T C1 = K^-1 k*
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B, Ppad, 0.0, C1,
                         Ppad, 3)) return rc;
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_A.as<double>(), Npad, C1, Ppad, 0.0, Z,
                         Ppad, 5)) return rc;
  if (acq_kind >= TGP_ACQ_MES && h->ent_S == 0)
    return fail(h, TGP_ERR_STATE, "entropy-search acquisition needs min-value samples: call tgp_set_min_value_samples");
  tgp_handle twin = acq_kind == TGP_ACQ_GIBBON ? h->rep_twin : nullptr;
  launch_grad_tail(h->stream, m, dXq, P, Ppad, B, C1, Z, gpart, acq_kind, param, dval, dgrad, h->d_ent.as<double>(), h->ent_S,
                   twin ? h->rep_weight : 0.0, 0.0);
  if (twin) {  // + w/2 log(var_twin + noise): the same pipeline on the conditioned model, accumulated
    if (!twin->have_data) return fail(h, TGP_ERR_STATE, "the repulsion twin has no data");
    const int64_t Nt = twin->Npad;
    double* Bt = h->s_grad.as<double>();
    double* C1t = Bt + (size_t)Nt * Ppad;
    double* Zt = C1t + (size_t)Nt * Ppad;
    const ModelDev mt = model_dev(twin);
    launch_kstar_t(h->stream, mt, dXq, P, Ppad, Bt);
    if (int rc = gemm_tall(h, false, (int)Nt, (int)Ppad, (int)Nt, 1.0, twin->d_W.as<double>(), Nt, Bt, Ppad, 0.0, C1t,
                           Ppad, 3)) return rc;
    if (int rc = gemm_tall(h, false, (int)Nt, (int)Ppad, (int)Nt, 1.0, twin->d_A.as<double>(), Nt, C1t, Ppad, 0.0, Zt,
                           Ppad, 5)) return rc;
    launch_grad_tail(h->stream, mt, dXq, P, Ppad, Bt, C1t, Zt, gpart, ACQ_LOGYVAR, 0.0, dval, dgrad, nullptr, 0, 0.0,
                     0.5 * h->rep_weight);
  }
  if (h->pen_kind != 0 && h->pen_P > 0) {
    const double* pend = h->d_pen.as<double>();
    launch_penalize_grad(h->stream, dval, dgrad, dXq, P, h->d, h->pen_kind, h->pen_P, pend,
                         pend + (size_t)h->pen_P * h->d, pend + (size_t)h->pen_P * (h->d + 1));
  }
  if (int rc = stage_out_finish(h, dval, val, P, where)) return rc;
  if (int rc = stage_out_finish(h, dgrad, grad, (size_t)P * h->d, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

// ---- the handful of q-batches of an L-BFGS-B iteration over a BATCH acquisition function (qEI; round 6) ------------------------
// The reference optimises BatchMonteCarloExpectedImprovement with batchify_joint(generate_continuous_optimizer) (optimizer.py:
// 897-934, 344-560): every iterate of every start is a group of q points, value and gradient by TF autodiff THROUGH predict_joint
// (interface.py:126-133), the Cholesky factor of the q x q covariance and the reparametrised samples (sampler.py:276-287).  Here:
// tgp_joint_forward = predict_joint of those few groups in the skinny-product form of tgp_predict at a handful of points (the joint
// kernel is built for 10^5 groups: one 250-point block walks all of W, 5 ms at N = 2048), tgp_joint_vjp = the vector-Jacobian
// product of predict_joint (tgp_kernels_grad.hip); the q x q factorisation and its adjoint in between are host arithmetic.
constexpr int64_t JOINT_SMALL_P = 2048;   // points per call (groups x q)
static int joint_small_common(tgp_handle h, const double* Xq, int64_t G, int q, int where, const double** dXq, int64_t* P,
                              int64_t* Ppad) {
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (G < 1 || q < 1 || !Xq) return fail(h, TGP_ERR_SHAPE, "need G >= 1 groups of q >= 1 points");
  if (G * (int64_t)q > JOINT_SMALL_P)
    return fail(h, TGP_ERR_SHAPE, "G * q = %lld points: at most %lld per call (chunk the groups)", (long long)(G * q),
                (long long)JOINT_SMALL_P);
  if (int rc = set_device(h)) return rc;
  *P = G * (int64_t)q;
  *Ppad = ((*P + 63) / 64) * 64;
  return stage_in(h, h->s_in, Xq, (size_t)*P * h->d, where, dXq);
}

// mean [P], cov [P][q] (device) of P = G q resident points: K*^T, W K*, ONE Gram product, the small-predict tail, the pick kernel
static int joint_small_device(tgp_handle h, const double* dXq, int64_t P, int64_t Ppad, int q, double* dmean, double* dcov) {
  const int64_t Npad = h->Npad;
  HIPCHK(h, h->s_grad.reserve(((size_t)3 * Npad * Ppad + (size_t)Ppad * Ppad + predict_small_scratch_doubles(Ppad)) *
                              sizeof(double)));
  double* B = h->s_grad.as<double>();
  double* C1 = B + (size_t)Npad * Ppad;
  double* C1t = C1 + (size_t)Npad * Ppad;
  double* S = C1t + (size_t)Npad * Ppad;
  double* part = S + (size_t)Ppad * Ppad;
  const ModelDev m = model_dev(h);
  launch_kstar_t(h->stream, m, dXq, P, Ppad, B);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B, Ppad, 0.0, C1, Ppad, 3))
    return rc;
  launch_transpose(h->stream, C1, Npad, Ppad, Ppad, C1t, Npad);
  if (int rc = gemm_tall(h, false, (int)Ppad, (int)Ppad, (int)Npad, 1.0, C1t, Npad, C1, Ppad, 0.0, S, Ppad, 0)) return rc;
  launch_predict_small_tail(h->stream, m, P, Ppad, B, C1, part, dmean, nullptr);
  launch_joint_pick(h->stream, m, dXq, P, Ppad, q, S, dcov);
  return TGP_OK;
}

int tgp_joint_forward(tgp_handle h, const double* Xq, int64_t G, int q, double* mean, double* cov, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!mean || !cov) return fail(h, TGP_ERR_ARG, "mean / cov is NULL");
  const double* dXq;
  int64_t P, Ppad;
  if (int rc = joint_small_common(h, Xq, G, q, where, &dXq, &P, &Ppad)) return rc;
  double *dmean, *dcov;
  if (int rc = stage_out_prepare(h, h->s_out1, mean, (size_t)P, where, &dmean)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out2, cov, (size_t)P * q, where, &dcov)) return rc;
  if (int rc = joint_small_device(h, dXq, P, Ppad, q, dmean, dcov)) return rc;
  if (int rc = stage_out_finish(h, dmean, mean, (size_t)P, where)) return rc;
  if (int rc = stage_out_finish(h, dcov, cov, (size_t)P * q, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_joint_vjp(tgp_handle h, const double* Xq, int64_t G, int q, const double* gmean, const double* gcov, double* grad,
                  int where) {
  if (!h) return TGP_ERR_ARG;
  if (!gmean || !gcov || !grad) return fail(h, TGP_ERR_ARG, "gmean / gcov / grad is NULL");
  const double* dXq;
  int64_t P, Ppad;
  if (int rc = joint_small_common(h, Xq, G, q, where, &dXq, &P, &Ppad)) return rc;
  const int64_t Npad = h->Npad;
  const double *dgm, *dgc;
  double* dgrad;
  if (int rc = stage_in(h, h->s_in2, gmean, (size_t)P, where, &dgm)) return rc;
  if (int rc = stage_in(h, h->s_out2, gcov, (size_t)P * q, where, &dgc)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, grad, (size_t)P * h->d, where, &dgrad)) return rc;
  HIPCHK(h, h->s_grad.reserve(((size_t)4 * Npad * Ppad + grad_tail_scratch_doubles(Ppad)) * sizeof(double)));
  double* B = h->s_grad.as<double>();
  double* C1 = B + (size_t)Npad * Ppad;
  double* D = C1 + (size_t)Npad * Ppad;
  double* Z = D + (size_t)Npad * Ppad;
  double* part = Z + (size_t)Npad * Ppad;
  const ModelDev m = model_dev(h);
  launch_kstar_t(h->stream, m, dXq, P, Ppad, B);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B, Ppad, 0.0, C1, Ppad, 3))
    return rc;
  launch_joint_mix(h->stream, C1, dgc, P, Ppad, Npad, q, D);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_A.as<double>(), Npad, D, Ppad, 0.0, Z, Ppad, 5))
    return rc;
  launch_joint_vjp_tail(h->stream, m, dXq, P, Ppad, q, B, C1, Z, part, dgm, dgc, dgrad);
  if (int rc = stage_out_finish(h, dgrad, grad, (size_t)P * h->d, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

// qEI value and gradient of a handful of q-batches in ONE call, everything on the device: tgp_joint_forward's arrays -> the
// one-wave-per-group tail (factorisation, samples, value, the adjoints of mean and covariance: qei_grad_tail_kernel) -> tgp_joint_vjp's
// second half on the SAME K*^T and W K*.  One host synchronisation.  q <= 64, G * q <= 2048, the tail's LDS holds S sample slots.
int tgp_qei_value_grad(tgp_handle h, const double* Xq, int64_t G, int q, const double* eps, int S, double eta, double jitter,
                       double* val, double* grad, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!(jitter >= 0.0)) return fail(h, TGP_ERR_ARG, "jitter must be >= 0");
  if (S < 1 || !eps) return fail(h, TGP_ERR_ARG, "need S >= 1 draws");
  if (!val || !grad) return fail(h, TGP_ERR_ARG, "val / grad is NULL");
  if (q > MAX_Q) return fail(h, TGP_ERR_SHAPE, "q must be in 1..%d, got %d", MAX_Q, q);
  if (q >= 1 && qei_grad_tail_lds_bytes(q, S) > (size_t)160 * 1024)
    return fail(h, TGP_ERR_SHAPE, "q = %d with S = %d draws does not fit the tail's LDS (tgp_joint_forward / tgp_joint_vjp take any S)", q, S);
  const double* dXq;
  int64_t P, Ppad;
  if (int rc = joint_small_common(h, Xq, G, q, where, &dXq, &P, &Ppad)) return rc;
  const int64_t Npad = h->Npad;
  const double* deps;
  double *dval, *dgrad;
  if (int rc = stage_in(h, h->s_in2, eps, (size_t)q * S, where, &deps)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, val, (size_t)G, where, &dval)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out2, grad, (size_t)P * h->d, where, &dgrad)) return rc;
  const size_t part_doubles = std::max(predict_small_scratch_doubles(Ppad), grad_tail_scratch_doubles(Ppad));
  HIPCHK(h, h->s_grad.reserve(((size_t)4 * Npad * Ppad + (size_t)Ppad * Ppad + part_doubles + (size_t)2 * P * (1 + q)) * sizeof(double)));
  double* B = h->s_grad.as<double>();
  double* C1 = B + (size_t)Npad * Ppad;
  double* T1 = C1 + (size_t)Npad * Ppad;   // C1^T, then D
  double* Z = T1 + (size_t)Npad * Ppad;
  double* Spp = Z + (size_t)Npad * Ppad;
  double* part = Spp + (size_t)Ppad * Ppad;
  double* dmean = part + part_doubles;
  double* dcov = dmean + P;
  double* dgm = dcov + (size_t)P * q;
  double* dgc = dgm + P;
  const ModelDev m = model_dev(h);
  HIPCHK(h, h->d_info.reserve(sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->d_info.p, 0, sizeof(int), h->stream));
  launch_kstar_t(h->stream, m, dXq, P, Ppad, B);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B, Ppad, 0.0, C1, Ppad, 3))
    return rc;
  launch_transpose(h->stream, C1, Npad, Ppad, Ppad, T1, Npad);
  if (int rc = gemm_tall(h, false, (int)Ppad, (int)Ppad, (int)Npad, 1.0, T1, Npad, C1, Ppad, 0.0, Spp, Ppad, 0)) return rc;
  launch_predict_small_tail(h->stream, m, P, Ppad, B, C1, part, dmean, nullptr);
  launch_joint_pick(h->stream, m, dXq, P, Ppad, q, Spp, dcov);
  launch_qei_grad_tail(h->stream, dmean, dcov, G, q, deps, S, eta, jitter, dval, dgm, dgc, h->d_info.as<int>());
  launch_joint_mix(h->stream, C1, dgc, P, Ppad, Npad, q, T1);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)Ppad, (int)Npad, 1.0, h->d_A.as<double>(), Npad, T1, Ppad, 0.0, Z, Ppad, 5))
    return rc;
  launch_joint_vjp_tail(h->stream, m, dXq, P, Ppad, q, B, C1, Z, part, dgm, dgc, dgrad);
  if (int rc = stage_out_finish(h, dval, val, (size_t)G, where)) return rc;
  if (int rc = stage_out_finish(h, dgrad, grad, (size_t)P * h->d, where)) return rc;
  if (int rc = sync(h)) return rc;
  int info = 0;
  HIPCHK(h, hipMemcpy(&info, h->d_info.p, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(h, hipGetLastError());
  if (info != 0)
    return fail(h, TGP_ERR_NOT_PD, "qEI: cov + jitter*I not positive definite for group %d", info - 1);
  return TGP_OK;
}

int tgp_cov_between(tgp_handle h, const double* X1, int64_t P1, const double* X2, int64_t P2, double* out,
                    int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (P1 < 0 || P2 < 0 || (P1 > 0 && P2 > 0 && (!X1 || !X2 || !out))) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (P1 == 0 || P2 == 0) return TGP_OK;
  if (int rc = set_device(h)) return rc;
  const int64_t P1p = ((P1 + 63) / 64) * 64, P2p = ((P2 + 63) / 64) * 64, Npad = h->Npad;
  const double *d1, *d2;
  double* dout;
  if (int rc = stage_in(h, h->s_in, X1, (size_t)P1 * h->d, where, &d1)) return rc;
  if (int rc = stage_in(h, h->s_in2, X2, (size_t)P2 * h->d, where, &d2)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, out, (size_t)P1 * P2, where, &dout)) return rc;
  HIPCHK(h, h->s_grad.reserve(((size_t)Npad * (3 * P1p + 2 * P2p) + (size_t)P1p * P2p) * sizeof(double)));
  double* B1 = h->s_grad.as<double>();
  double* C1 = B1 + (size_t)Npad * P1p;
  double* C1t = C1 + (size_t)Npad * P1p;
  double* B2 = C1t + (size_t)Npad * P1p;
  double* C2 = B2 + (size_t)Npad * P2p;
  double* S = C2 + (size_t)Npad * P2p;
  const ModelDev m = model_dev(h);
  hipStream_t s = h->stream;
  // A_i = L^-1 K(X, X_i) = W B_i;  S = A_1^T A_2  (the reference's two triangular solves + einsum)
  launch_kstar_t(s, m, d1, P1, P1p, B1);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)P1p, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B1, P1p, 0.0, C1, P1p, 3)) return rc;
  launch_kstar_t(s, m, d2, P2, P2p, B2);
  if (int rc = gemm_tall(h, false, (int)Npad, (int)P2p, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B2, P2p, 0.0, C2, P2p, 3)) return rc;
  launch_transpose(s, C1, Npad, P1p, P1p, C1t, Npad);
  if (int rc = gemm_tall(h, false, (int)P1p, (int)P2p, (int)Npad, 1.0, C1t, Npad, C2, P2p, 0.0, S, P2p, 0)) return rc;
  launch_cov_tail(s, m, d1, P1, d2, P2, S, P2p, dout);
  if (int rc = stage_out_finish(h, dout, out, (size_t)P1 * P2, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_sample_joint(tgp_handle h, const double* Xq, int64_t n, const double* eps, int S, double jitter,
                     double* out, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (n < 1 || S < 1 || !Xq || !eps || !out) return fail(h, TGP_ERR_SHAPE, "need n >= 1 points and S >= 1 draws");
  if (!(jitter >= 0.0)) return fail(h, TGP_ERR_ARG, "jitter must be non-negative");
  if (int rc = set_device(h)) return rc;
  const int64_t Pp = ((n + 63) / 64) * 64, Sp = (((int64_t)S + 63) / 64) * 64, Npad = h->Npad;
  const double *dXq, *deps;
  double* dout;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)n * h->d, where, &dXq)) return rc;
  if (int rc = stage_in(h, h->s_in2, eps, (size_t)n * S, where, &deps)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, out, (size_t)S * n, where, &dout)) return rc;
  const size_t need = (size_t)Npad * Pp * 3 + (size_t)Pp * Pp * 3 + (size_t)Pp * Sp * 2 + (size_t)Pp + 64;
  HIPCHK(h, h->s_grad.reserve(need * sizeof(double)));
  double* B1 = h->s_grad.as<double>();
  double* C1 = B1 + (size_t)Npad * Pp;
  double* C1t = C1 + (size_t)Npad * Pp;
  double* A = C1t + (size_t)Npad * Pp;  // [Pp, Pp] covariance, then scratch of the factorisation
  double* L = A + (size_t)Pp * Pp;
  double* W = L + (size_t)Pp * Pp;
  double* E = W + (size_t)Pp * Pp;      // [Pp, Sp] padded draws
  double* R = E + (size_t)Pp * Sp;      // [Pp, Sp] L E
  double* mean = R + (size_t)Pp * Sp;   // [Pp]
  int* info = (int*)(mean + Pp);
  const ModelDev m = model_dev(h);
  hipStream_t s = h->stream;
  HIPCHK(h, hipMemsetAsync(info, 0, sizeof(int), s));
  launch_predict_mean(s, m, dXq, n, mean);
  // cov = k(Xq, Xq) - (W k(X, Xq))^T (W k(X, Xq)) + jitter I   (gpflow predict_f(full_cov) + sample_mvn)
  launch_kstar_t(s, m, dXq, n, Pp, B1);
  launch_gemm(s, false, (int)Npad, (int)Pp, (int)Npad, 1.0, h->d_W.as<double>(), Npad, B1, Pp, 0.0, C1, Pp, false, 3);
  launch_transpose(s, C1, Npad, Pp, Pp, C1t, Npad);
  launch_gemm(s, false, (int)Pp, (int)Pp, (int)Npad, 1.0, C1t, Npad, C1, Pp, 0.0, L, Pp, true, 0);  // S (lower) in L
  launch_cov_sym_tail(s, m, dXq, n, Pp, L, jitter, A);
  HIPCHK(h, hipMemsetAsync(L, 0, (size_t)Pp * Pp * sizeof(double), s));
  HIPCHK(h, hipMemsetAsync(W, 0, (size_t)Pp * Pp * sizeof(double), s));
  chol_inv(s, FactorWs{A, L, W, Pp, info}, 0, Pp);
  launch_pad_copy(s, deps, n, S, E, Pp, Sp);
  launch_gemm(s, false, (int)Pp, (int)Sp, (int)Pp, 1.0, L, Pp, E, Sp, 0.0, R, Sp, false, 3);
  launch_sample_tail(s, mean, R, n, S, Sp, dout);
  int hinfo = 0;
  HIPCHK(h, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, s));
  if (int rc = stage_out_finish(h, dout, out, (size_t)S * n, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  if (hinfo != 0)
    return fail(h, TGP_ERR_NOT_PD, "Cholesky of the joint posterior covariance failed at point %d: increase the jitter",
                hinfo - 1);
  return TGP_OK;
}

// The k best (value desc, index asc) of device-resident values into device slots fv [k], fi [k]: enqueue only.
static int enqueue_topk_of_values(tgp_handle h, const double* dvals, int64_t M, int64_t index_base, int k, double* fv,
                                  int64_t* fi) {
  HIPCHK(h, h->s_blkv.reserve((size_t)std::max(512, 8 * k) * sizeof(double)));
  HIPCHK(h, h->s_blki.reserve((size_t)std::max(512, 8 * k) * sizeof(int64_t)));
  if (M <= topk_small_max()) {
    launch_topk_small(h->stream, dvals, M, index_base, k, fv, fi, h->s_blkv.as<double>(), h->s_blki.as<int64_t>());
  } else {
    for (int t = 0; t < k; ++t)  // thresholds stay on the device: no host round trip per pass
      launch_topk_pass(h->stream, dvals, M, index_base, t ? fv + t - 1 : nullptr, t ? fi + t - 1 : nullptr,
                       h->s_blkv.as<double>(), h->s_blki.as<int64_t>(), fv + t, fi + t);
  }
  return TGP_OK;
}

// Fused predict + acquisition + arg-max over device-resident candidates: the winner's value and global index go
// to the DEVICE slots dval / didx; nothing is synchronised.  The posterior tails run the fused sweep (values never
// reach HBM); penalised / entropy tails take one trip through HBM (8 B per candidate) between sweep and arg-max.
static int enqueue_argmax(tgp_handle h, int acq_kind, double param, const double* dXq, int64_t M, int64_t index_base,
                          double* dval, int64_t* didx) {
  h->last_launches = 0;
  h->last_ms = 0.0;
  if ((h->pen_kind != 0 && h->pen_P > 0) || acq_kind >= TGP_ACQ_MES) {
    HIPCHK(h, h->s_out3.reserve((size_t)M * sizeof(double)));
    double* dvals = h->s_out3.as<double>();
    if (int rc = acq_values_device(h, acq_kind, param, dXq, M, dvals)) return rc;
    return enqueue_topk_of_values(h, dvals, M, index_base, 1, dval, didx);
  }
  SweepArgs a{};
  a.m = model_dev(h);
  a.Xq = dXq;
  a.M = M;
  a.acq_kind = acq_kind;
  a.acq_param = param;
  a.index_base = index_base;
  const int64_t grid = sweep_blocks(h, a, false);
  HIPCHK(h, h->s_blkv.reserve(grid * sizeof(double)));
  HIPCHK(h, h->s_blki.reserve(grid * sizeof(int64_t)));
  a.blk_val = h->s_blkv.as<double>();
  a.blk_idx = h->s_blki.as<int64_t>();
  HIPCHK(h, launch_sweep_timed(h, a, false));
  launch_argmax_final(h->stream, a.blk_val, a.blk_idx, grid, dval, didx);
  return TGP_OK;
}

static int argmax_checks(tgp_handle h, int acq_kind, const double* Xq, int64_t M) {
  if (!h) return TGP_ERR_ARG;
  if (acq_kind < 0 || acq_kind > ACQ_KIND_MAX) return fail(h, TGP_ERR_ARG, "unknown acquisition kind %d", acq_kind);
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (M < 1) return fail(h, TGP_ERR_SHAPE, "arg-max over an empty candidate set");
  if (!Xq) return fail(h, TGP_ERR_ARG, "Xq is NULL");
  if (acq_kind >= TGP_ACQ_MES && h->ent_S == 0)
    return fail(h, TGP_ERR_STATE, "entropy-search acquisition needs min-value samples: call tgp_set_min_value_samples");
  return TGP_OK;
}

int tgp_acq_argmax(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t M,
                   int64_t index_base, double* best_val, int64_t* best_idx, double* best_x, int where) {
  if (int rc = argmax_checks(h, acq_kind, Xq, M)) return rc;
  if (int rc = set_device(h)) return rc;
  const double* dXq;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
  HIPCHK(h, h->s_small.reserve(64));
  double* fv = h->s_small.as<double>();
  int64_t* fi = (int64_t*)(fv + 1);
  double hv;
  int64_t hi;
  for (int attempt = 0;; ++attempt) {
    if (int rc = enqueue_argmax(h, acq_kind, param, dXq, M, index_base, fv, fi)) return rc;
    HIPCHK(h, hipMemcpyAsync(&hv, fv, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&hi, fi, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    if (int rc = sync(h)) return rc;
    HIPCHK(h, hipGetLastError());
    if (attempt < 2 && auto_canary_tripped(h)) continue;  // TGP_PREC_AUTO: a sample broke its bound -> next rung, again
    break;
  }
  if (best_val) *best_val = hv;
  if (best_idx) *best_idx = hi;
  if (best_x) {
    const int64_t local = hi - index_base;
    if (local < 0 || local >= M) return fail(h, TGP_ERR_HIP, "arg-max produced no valid index (all NaN?)");
    if (where == TGP_DEVICE)
      HIPCHK(h, hipMemcpy(best_x, dXq + local * h->d, h->d * sizeof(double), hipMemcpyDeviceToHost));
    else
      memcpy(best_x, Xq + local * h->d, h->d * sizeof(double));
  }
  return TGP_OK;
}

int tgp_acq_argmax_async(tgp_handle h, int acq_kind, double param, const double* Xq_device, int64_t M,
                         int64_t index_base, double* pair_device) {
  if (int rc = argmax_checks(h, acq_kind, Xq_device, M)) return rc;
  if (!pair_device) return fail(h, TGP_ERR_ARG, "pair_device is NULL");
  if (int rc = set_device(h)) return rc;
  if (int rc = enqueue_argmax(h, acq_kind, param, Xq_device, M, index_base, pair_device, (int64_t*)(pair_device + 1)))
    return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_merge_winners_async(tgp_handle h, const double* gathered_device, int P, int V, int minimize,
                            double* out_device) {
  if (!h) return TGP_ERR_ARG;
  if (P < 1 || V < 1 || !gathered_device || !out_device) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (int rc = set_device(h)) return rc;
  launch_merge_winners(h->stream, gathered_device, P, V, minimize ? 1 : 0, out_device);
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_acq_topk(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t M,
                 int64_t index_base, int k, double* vals, int64_t* idx, int where) {
  if (!h) return TGP_ERR_ARG;
  if (k < 1 || k > 1024) return fail(h, TGP_ERR_ARG, "k must be in 1..1024");
  if (M < k) return fail(h, TGP_ERR_SHAPE, "top-k needs M >= k (M=%lld, k=%d)", (long long)M, k);
  if (acq_kind < 0 || acq_kind > ACQ_KIND_MAX) return fail(h, TGP_ERR_ARG, "unknown acquisition kind %d", acq_kind);
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (int rc = set_device(h)) return rc;
  // acquisition values stay on the device (8 B / candidate), then k extraction passes
  HIPCHK(h, h->s_out3.reserve((size_t)M * sizeof(double)));
  double* dvals = h->s_out3.as<double>();
  const double* dXq;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
  HIPCHK(h, h->s_small.reserve((size_t)k * 16 + 64));
  double* fv = h->s_small.as<double>();       // [k] winners' values
  int64_t* fi = (int64_t*)(fv + k);           // [k] winners' indices
  for (int attempt = 0;; ++attempt) {
    if (int rc = acq_values_device(h, acq_kind, param, dXq, M, dvals)) return rc;
    if (int rc = enqueue_topk_of_values(h, dvals, M, index_base, k, fv, fi)) return rc;
    HIPCHK(h, hipMemcpyAsync(vals, fv, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(idx, fi, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    if (int rc = sync(h)) return rc;
    HIPCHK(h, hipGetLastError());
    if (attempt < 2 && auto_canary_tripped(h)) continue;
    break;
  }
  return TGP_OK;
}

int tgp_sample_box(tgp_handle h, uint64_t seed, int64_t first, int64_t M, const double* lower,
                   const double* upper, double* out_device) {
  if (!h) return TGP_ERR_ARG;
  if (M < 0 || !lower || !upper || (M > 0 && !out_device)) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (M == 0) return TGP_OK;
  if (int rc = set_device(h)) return rc;
  HIPCHK(h, h->s_small.reserve(2 * MAX_D * sizeof(double) + 64));
  double* dl = h->s_small.as<double>() + 8;
  double* du = dl + MAX_D;
  HIPCHK(h, hipMemcpyAsync(dl, lower, h->d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(du, upper, h->d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  launch_sample_box(h->stream, seed, first, M, h->d, dl, du, out_device);
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

static int joint_common(tgp_handle h, const double* Xq, int64_t G, int q, int where, const double** dXq,
                        double** dmean, double** dcov, double* mean, double* cov, bool force_dev_out) {
  if (!h) return TGP_ERR_ARG;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (q < 1 || q > MAX_Q) return fail(h, TGP_ERR_SHAPE, "q must be in 1..%d, got %d", MAX_Q, q);
  if (G < 0 || (G > 0 && !Xq)) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (int rc = set_device(h)) return rc;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)G * q * h->d, where, dXq)) return rc;
  if (force_dev_out) {
    HIPCHK(h, h->s_out1.reserve((size_t)G * q * sizeof(double)));
    HIPCHK(h, h->s_out2.reserve((size_t)G * q * q * sizeof(double)));
    *dmean = h->s_out1.as<double>();
    *dcov = h->s_out2.as<double>();
  } else {
    if (int rc = stage_out_prepare(h, h->s_out1, mean, (size_t)G * q, where, dmean)) return rc;
    if (int rc = stage_out_prepare(h, h->s_out2, cov, (size_t)G * q * q, where, dcov)) return rc;
  }
  // a handful of groups (round 6): the skinny-product form of tgp_joint_forward -- the joint kernel is built for 10^5 groups, ONE
  // of its 250-point blocks walks all of W (4.9 ms at N = 2048, 18 ms at N = 4096 whatever the count); variant bit 10 or a joint
  // policy bit keep the kernel
  // (whatever tgp_set_precision says: joint mode is float64 on every path)
  if (G * (int64_t)q <= JOINT_SMALL_P && !(h->variant & (VARIANT_SWEEP_SMALL_PREDICT | VARIANT_JOINT_V1))) {
    const int64_t P = G * (int64_t)q, Ppad = ((P + 63) / 64) * 64;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));   // (the callers read the elapsed time between the two events)
    if (int rc = joint_small_device(h, *dXq, P, Ppad, q, *dmean, *dcov)) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->last_launches = 0;
    return TGP_OK;
  }
  SweepArgs a{};
  a.m = model_dev(h);
  a.Xq = *dXq;
  a.M = G * q;
  a.G = G;
  a.q = q;
  a.mean_out = *dmean;
  a.cov_out = *dcov;
  a.acq_kind = -1;
  HIPCHK(h, launch_sweep_timed(h, a, true));
  return TGP_OK;
}

int tgp_predict_joint(tgp_handle h, const double* Xq, int64_t G, int q, double* mean, double* cov,
                      int where) {
  const double* dXq;
  double *dmean, *dcov;
  if (G == 0) return h ? TGP_OK : TGP_ERR_ARG;
  if (int rc = joint_common(h, Xq, G, q, where, &dXq, &dmean, &dcov, mean, cov, false)) return rc;
  if (int rc = stage_out_finish(h, dmean, mean, (size_t)G * q, where)) return rc;
  if (int rc = stage_out_finish(h, dcov, cov, (size_t)G * q * q, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_qei(tgp_handle h, const double* Xq, int64_t G, int q, const double* eps, int S, double eta,
            double jitter, double* out, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!(jitter >= 0.0)) return fail(h, TGP_ERR_ARG, "jitter must be >= 0");
  if (S < 1 || !eps) return fail(h, TGP_ERR_ARG, "need S >= 1 draws");
  if (G == 0) return TGP_OK;
  if (!out) return fail(h, TGP_ERR_ARG, "out is NULL");
  const double* dXq;
  double *dmean, *dcov;
  // chunk the groups so the [G,q,q] covariances stay within a bounded scratch (<= ~1 GiB)
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(1ull << 30) / ((int64_t)q * q * 8));
  if (int rc = set_device(h)) return rc;
  const double* deps;
  if (int rc = stage_in(h, h->s_in2, eps, (size_t)q * S, where, &deps)) return rc;
  HIPCHK(h, h->d_info.reserve(sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->d_info.p, 0, sizeof(int), h->stream));
  double total_ms = 0.0;
  int launches = 0;
  for (int64_t g0 = 0; g0 < G; g0 += chunk) {
    const int64_t gc = std::min(chunk, G - g0);
    if (int rc = joint_common(h, Xq + g0 * q * h->d, gc, q, where, &dXq, &dmean, &dcov, nullptr, nullptr, true))
      return rc;
    double* dout;
    if (int rc = stage_out_prepare(h, h->s_out3, out + g0, gc, where, &dout)) return rc;
    launch_qei_tail(h->stream, dmean, dcov, gc, q, deps, S, eta, jitter, dout, nullptr, h->d_info.as<int>());
    if (int rc = stage_out_finish(h, dout, out + g0, gc, where)) return rc;
    if (int rc = sync(h)) return rc;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) total_ms += ms;
    ++launches;
  }
  h->last_ms = total_ms;
  h->last_launches = launches;
  int info = 0;
  HIPCHK(h, hipMemcpy(&info, h->d_info.p, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(h, hipGetLastError());
  if (info != 0)
    return fail(h, TGP_ERR_NOT_PD, "qEI: cov + jitter*I not positive definite for group %d", info - 1);
  return TGP_OK;
}

int tgp_reparam_samples(tgp_handle h, const double* Xq, int64_t G, int q, const double* eps, int S,
                        double jitter, double* out, int where) {
  if (!h) return TGP_ERR_ARG;
  if (!(jitter >= 0.0)) return fail(h, TGP_ERR_ARG, "jitter must be >= 0");
  if (S < 1 || !eps) return fail(h, TGP_ERR_ARG, "need S >= 1 draws");
  if (G == 0) return TGP_OK;
  if (!out) return fail(h, TGP_ERR_ARG, "out is NULL");
  const double* dXq;
  double *dmean, *dcov;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(1ull << 28) / ((int64_t)q * std::max(q, S) * 8));
  if (int rc = set_device(h)) return rc;
  const double* deps;
  if (int rc = stage_in(h, h->s_in2, eps, (size_t)q * S, where, &deps)) return rc;
  HIPCHK(h, h->d_info.reserve(sizeof(int)));
  HIPCHK(h, hipMemsetAsync(h->d_info.p, 0, sizeof(int), h->stream));
  for (int64_t g0 = 0; g0 < G; g0 += chunk) {
    const int64_t gc = std::min(chunk, G - g0);
    if (int rc = joint_common(h, Xq + g0 * q * h->d, gc, q, where, &dXq, &dmean, &dcov, nullptr, nullptr, true))
      return rc;
    double* dout;
    if (int rc = stage_out_prepare(h, h->s_out3, out + g0 * S * q, (size_t)gc * S * q, where, &dout)) return rc;
    launch_qei_tail(h->stream, dmean, dcov, gc, q, deps, S, 0.0, jitter, nullptr, dout, h->d_info.as<int>());
    if (int rc = stage_out_finish(h, dout, out + g0 * S * q, (size_t)gc * S * q, where)) return rc;
    if (int rc = sync(h)) return rc;
  }
  int info = 0;
  HIPCHK(h, hipMemcpy(&info, h->d_info.p, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(h, hipGetLastError());
  if (info != 0)
    return fail(h, TGP_ERR_NOT_PD, "reparam samples: cov + jitter*I not positive definite for group %d", info - 1);
  return TGP_OK;
}

// ---- trajectories -----------------------------------------------------------------------------
static TrajDev traj_dev(tgp_traj t) {
  TrajDev td;
  td.m = model_dev(t->h);
  td.F = t->F;
  td.B = t->B;
  td.rffW = t->d_W.as<double>();
  td.rffb = t->d_b.as<double>();
  td.rffW_ht = td.rffW + (size_t)t->F * t->h->dp;
  td.rffb_ht = td.rffb + t->F;
  td.ws = t->d_ws.as<double>();
  td.v = t->d_v.as<double>();
  td.canonical = t->canonical;
  return td;
}

int tgp_traj_create(tgp_handle h, const double* rff_W, const double* rff_b, int F, const double* w,
                    const double* xi, int B, tgp_traj* out) {
  if (!h || !out) return TGP_ERR_ARG;
  *out = nullptr;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (F < 1 || B < 1 || !rff_W || !rff_b || !w || !xi) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (int rc = set_device(h)) return rc;
  tgp_traj t = new (std::nothrow) tgp_traj_s();
  if (!t) return fail(h, TGP_ERR_ALLOC, "host allocation failed");
  t->h = h;
  t->device = h->device;
  t->F = F;
  t->B = B;
  const int d = h->d, dp = h->dp;
  const int64_t N = h->N, Npad = h->Npad;
  std::vector<double> Wp((size_t)F * dp, 0.0), ws((size_t)F * B);
  for (int f = 0; f < F; ++f)
    for (int c = 0; c < d; ++c) Wp[(size_t)f * dp + c] = rff_W[(size_t)f * d + c];
  const double scale = std::sqrt(2.0 * h->variance / (double)F);
  for (size_t e = 0; e < (size_t)F * B; ++e) ws[e] = scale * w[e];
  // u = err + sqrt(noise) xi   (host, [N][B]) ; diff = u - Phi_Z w (device)
  std::vector<double> errh((size_t)N);
  hipError_t e;
#define TCHK(expr)                                                            \
  if ((e = (expr)) != hipSuccess) {                                           \
    tgp_traj_destroy(t);                                                      \
    return fail(h, e == hipErrorOutOfMemory ? TGP_ERR_ALLOC : TGP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); \
  }
  // the basis twice: as given (radians: features matrix, gradients) and in half turns for the evaluation kernel,
  // W / pi and b / pi + 1/2, so that cos(x . W + b) = sin(pi y) needs no range reduction by pi (tgp_kernels_traj.hip)
  TCHK(t->d_W.reserve(2 * Wp.size() * sizeof(double)));
  TCHK(t->d_b.reserve(2 * (size_t)F * sizeof(double)));
  TCHK(t->d_ws.reserve(ws.size() * sizeof(double)));
  TCHK(t->d_v.reserve((size_t)Npad * B * sizeof(double)));
  TCHK(hipMemcpy(t->d_W.p, Wp.data(), Wp.size() * sizeof(double), hipMemcpyHostToDevice));
  TCHK(hipMemcpy(t->d_b.p, rff_b, (size_t)F * sizeof(double), hipMemcpyHostToDevice));
  {
    constexpr double INV_PI = 0.31830988618379067154;
    std::vector<double> Wh(Wp.size()), bh((size_t)F);
    for (size_t e2 = 0; e2 < Wp.size(); ++e2) Wh[e2] = Wp[e2] * INV_PI;
    for (int f = 0; f < F; ++f) bh[f] = rff_b[f] * INV_PI + 0.5;
    TCHK(hipMemcpy(t->d_W.as<double>() + Wp.size(), Wh.data(), Wh.size() * sizeof(double), hipMemcpyHostToDevice));
    TCHK(hipMemcpy(t->d_b.as<double>() + F, bh.data(), bh.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  TCHK(hipMemcpy(t->d_ws.p, ws.data(), ws.size() * sizeof(double), hipMemcpyHostToDevice));
  TCHK(hipMemcpy(errh.data(), h->d_err.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
  // Phi_Z w  -> s_out1 [N][B]
  TCHK(h->s_out1.reserve((size_t)N * B * sizeof(double)));
  TrajDev td = traj_dev(t);
  launch_rff_project(h->stream, td, h->d_X.as<double>(), N, h->s_out1.as<double>());
  std::vector<double> proj((size_t)N * B);
  TCHK(hipMemcpyAsync(proj.data(), h->s_out1.p, proj.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  TCHK(hipStreamSynchronize(h->stream));
  // per trajectory: diff_b (padded) -> tmp1 ; tmp2 = W diff ; v_b = Wt tmp2
  const double sn = std::sqrt(h->noise);
  std::vector<double> diff((size_t)Npad, 0.0), vhost((size_t)Npad * B, 0.0), vb((size_t)Npad);
  for (int b = 0; b < B; ++b) {
    for (int64_t i = 0; i < N; ++i) diff[i] = errh[i] + sn * xi[i * B + b] - proj[i * B + b];
    TCHK(hipMemcpy(h->d_tmp1.p, diff.data(), (size_t)Npad * sizeof(double), hipMemcpyHostToDevice));
    launch_trmv(h->stream, h->d_W.as<double>(), Npad, Npad, h->d_tmp1.as<double>(), h->d_tmp2.as<double>(), true);
    launch_trmv(h->stream, h->d_A.as<double>(), Npad, Npad, h->d_tmp2.as<double>(), h->d_tmp1.as<double>(), false);
    TCHK(hipMemcpyAsync(vb.data(), h->d_tmp1.p, (size_t)Npad * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    TCHK(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < N; ++i) vhost[i * B + b] = vb[i];
  }
  TCHK(hipMemcpy(t->d_v.p, vhost.data(), vhost.size() * sizeof(double), hipMemcpyHostToDevice));
  TCHK(hipGetLastError());
#undef TCHK
  *out = t;
  return TGP_OK;
}

// == RandomFourierFeatureTrajectorySampler (sampler.py:452-591): posterior over the weights theta of the
// F scaled Fourier features given the data, in design space (F < N: F x F) or gram space (N <= F: N x N),
// theta = mean + chol(cov) eps; all matrices are built, factorised and applied on the device.
int tgp_traj_create_rff(tgp_handle h, const double* rff_W, const double* rff_b, int F, const double* eps, int B,
                        tgp_traj* out) {
  if (!h || !out) return TGP_ERR_ARG;
  *out = nullptr;
  if (!h->have_data) return fail(h, TGP_ERR_STATE, "model has no data: call tgp_set_data first");
  if (F < 1 || B < 1 || !rff_W || !rff_b || !eps) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (int rc = set_device(h)) return rc;
  tgp_traj t = new (std::nothrow) tgp_traj_s();
  if (!t) return fail(h, TGP_ERR_ALLOC, "host allocation failed");
  t->h = h;
  t->device = h->device;
  t->F = F;
  t->B = B;
  t->canonical = 0;
  const int d = h->d, dp = h->dp;
  const int64_t N = h->N, Npad = h->Npad;
  const int64_t Fp = (((int64_t)F + 63) / 64) * 64, Bp = (((int64_t)B + 63) / 64) * 64;
  std::vector<double> Wp((size_t)F * dp, 0.0);
  for (int f = 0; f < F; ++f)
    for (int c = 0; c < d; ++c) Wp[(size_t)f * dp + c] = rff_W[(size_t)f * d + c];
  const double scale = std::sqrt(2.0 * h->variance / (double)F);
  hipError_t e;
#define TCHK(expr)                                                            \
  if ((e = (expr)) != hipSuccess) {                                           \
    tgp_traj_destroy(t);                                                      \
    return fail(h, e == hipErrorOutOfMemory ? TGP_ERR_ALLOC : TGP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e)); \
  }
  // the basis twice: as given (radians: features matrix, gradients) and in half turns for the evaluation kernel,
  // W / pi and b / pi + 1/2, so that cos(x . W + b) = sin(pi y) needs no range reduction by pi (tgp_kernels_traj.hip)
  TCHK(t->d_W.reserve(2 * Wp.size() * sizeof(double)));
  TCHK(t->d_b.reserve(2 * (size_t)F * sizeof(double)));
  TCHK(t->d_ws.reserve((size_t)F * B * sizeof(double)));
  TCHK(t->d_theta.reserve((size_t)F * B * sizeof(double)));
  TCHK(t->d_v.reserve(64));
  TCHK(hipMemcpy(t->d_W.p, Wp.data(), Wp.size() * sizeof(double), hipMemcpyHostToDevice));
  TCHK(hipMemcpy(t->d_b.p, rff_b, (size_t)F * sizeof(double), hipMemcpyHostToDevice));
  {
    constexpr double INV_PI = 0.31830988618379067154;
    std::vector<double> Wh(Wp.size()), bh((size_t)F);
    for (size_t e2 = 0; e2 < Wp.size(); ++e2) Wh[e2] = Wp[e2] * INV_PI;
    for (int f = 0; f < F; ++f) bh[f] = rff_b[f] * INV_PI + 0.5;
    TCHK(hipMemcpy(t->d_W.as<double>() + Wp.size(), Wh.data(), Wh.size() * sizeof(double), hipMemcpyHostToDevice));
    TCHK(hipMemcpy(t->d_b.as<double>() + F, bh.data(), bh.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  const bool design = (int64_t)F < N;
  const int64_t Q = design ? Fp : Npad;  // side of the first factorisation
  // workspace: Phi, Phit [Npad Fp]; R0 [Npad 64]; X1, X2 [max 64]; E, RE [Fp Bp]; squares S1..S3 [Q Q]; C1..C3 [Fp Fp]
  const size_t nphi = (size_t)Npad * Fp, nsq = (size_t)Q * Q, nc = (size_t)Fp * Fp;
  const size_t need = 3 * nphi + (size_t)Npad * 64 * 2 + (size_t)Fp * 64 * 2 + 2 * (size_t)Fp * Bp + 3 * nsq + 3 * nc + 64;
  TCHK(h->s_grad.reserve(need * sizeof(double)));
  double* Phi = h->s_grad.as<double>();
  double* Phit = Phi + nphi;
  double* Aw = Phit + nphi;                 // gram space: A = W_G Phi, then its transpose reuses Phit
  double* R0 = Aw + nphi;                    // [Npad][64]: column 0 = Y - c
  double* Y1 = R0 + (size_t)Npad * 64;       // [Npad][64]
  double* M0 = Y1 + (size_t)Npad * 64;       // [Fp][64]
  double* M1 = M0 + (size_t)Fp * 64;         // [Fp][64]: column 0 = posterior mean
  double* E = M1 + (size_t)Fp * 64;          // [Fp][Bp]
  double* RE = E + (size_t)Fp * Bp;
  double* S1 = RE + (size_t)Fp * Bp;         // first factorisation: matrix / L / W
  double* S2 = S1 + nsq;
  double* S3 = S2 + nsq;
  double* C1 = S3 + nsq;                     // weight covariance: matrix / L / W
  double* C2 = C1 + nc;
  double* C3 = C2 + nc;
  int* info = (int*)(C3 + nc);
  hipStream_t s = h->stream;
  TrajDev td = traj_dev(t);
  TCHK(hipMemsetAsync(info, 0, 2 * sizeof(int), s));
  launch_rff_features(s, td, scale, Fp, Phi);
  launch_transpose(s, Phi, Npad, Fp, Fp, Phit, Npad);
  launch_pad_copy(s, h->d_err.as<double>(), N, 1, R0, Npad, 64);
  TCHK(h->s_in.reserve((size_t)F * B * sizeof(double)));
  TCHK(hipMemcpyAsync(h->s_in.p, eps, (size_t)F * B * sizeof(double), hipMemcpyHostToDevice, s));
  launch_pad_copy(s, h->s_in.as<double>(), F, B, E, Fp, Bp);
  TCHK(hipMemsetAsync(S2, 0, 2 * nsq * sizeof(double), s));
  TCHK(hipMemsetAsync(C2, 0, 2 * nc * sizeof(double), s));
  if (design) {
    // D = Phi^T Phi + noise I;  D^-1 = W_D^T W_D;  mean = D^-1 Phi^T r;  cov = noise D^-1   (sampler.py:529-556)
    launch_gemm(s, true, (int)Fp, (int)Fp, (int)Npad, 1.0, Phit, Npad, Phit, Npad, 0.0, S1, Fp, true, 0);
    launch_sym_finish(s, S1, F, Fp, 1.0, h->noise, 0);
    chol_inv(s, FactorWs{S1, S2, S3, Fp, info}, 0, Fp);
    launch_transpose(s, S3, Fp, Fp, Fp, S1, Fp);                                                   // W_D^T (S1 is dead)
    launch_gemm(s, false, (int)Fp, (int)Fp, (int)Fp, 1.0, S1, Fp, S3, Fp, 0.0, C1, Fp, false, 4);  // D^-1
    launch_gemm(s, false, (int)Fp, 64, (int)Npad, 1.0, Phit, Npad, R0, 64, 0.0, M0, 64, false, 0);  // Phi^T r
    launch_gemm(s, false, (int)Fp, 64, (int)Fp, 1.0, C1, Fp, M0, 64, 0.0, M1, 64, false, 0);        // mean
    launch_sym_finish(s, C1, F, Fp, h->noise, 0.0, 0);
  } else {
    // G = Phi Phi^T + noise I;  A = L_G^-1 Phi;  mean = A^T L_G^-1 r;  cov = I - A^T A            (sampler.py:558-591)
    launch_gemm(s, true, (int)Npad, (int)Npad, (int)Fp, 1.0, Phi, Fp, Phi, Fp, 0.0, S1, Npad, true, 0);
    launch_sym_finish(s, S1, N, Npad, 1.0, h->noise, 0);
    chol_inv(s, FactorWs{S1, S2, S3, Npad, info}, 0, Npad);
    launch_gemm(s, false, (int)Npad, (int)Fp, (int)Npad, 1.0, S3, Npad, Phi, Fp, 0.0, Aw, Fp, false, 3);
    launch_gemm(s, false, (int)Npad, 64, (int)Npad, 1.0, S3, Npad, R0, 64, 0.0, Y1, 64, false, 3);
    launch_transpose(s, Aw, Npad, Fp, Fp, Phit, Npad);                                              // A^T
    launch_gemm(s, false, (int)Fp, 64, (int)Npad, 1.0, Phit, Npad, Y1, 64, 0.0, M1, 64, false, 0);  // mean
    launch_gemm(s, true, (int)Fp, (int)Fp, (int)Npad, 1.0, Phit, Npad, Phit, Npad, 0.0, C1, Fp, true, 0);
    launch_sym_finish(s, C1, F, Fp, 1.0, 1.0, 1);
  }
  chol_inv(s, FactorWs{C1, C2, C3, Fp, info + 1}, 0, Fp);
  launch_gemm(s, false, (int)Fp, (int)Bp, (int)Fp, 1.0, C2, Fp, E, Bp, 0.0, RE, Bp, false, 3);
  launch_theta_tail(s, M1, 64, RE, Bp, F, B, scale, t->d_theta.as<double>(), t->d_ws.as<double>());
  int hinfo[2] = {0, 0};
  TCHK(hipMemcpyAsync(hinfo, info, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  TCHK(hipStreamSynchronize(s));
  TCHK(hipGetLastError());
#undef TCHK
  if (hinfo[0] != 0 || hinfo[1] != 0) {
    tgp_traj_destroy(t);
    return fail(h, TGP_ERR_NOT_PD, "Cholesky failed in the RFF weight posterior (%s, pivot %d)",
                hinfo[0] ? (design ? "design matrix" : "gram matrix") : "weight covariance",
                (hinfo[0] ? hinfo[0] : hinfo[1]) - 1);
  }
  *out = t;
  return TGP_OK;
}

int tgp_traj_get_theta(tgp_traj t, double* theta) {
  if (!t || !theta) return TGP_ERR_ARG;
  tgp_handle h = t->h;
  if (t->canonical) return fail(h, TGP_ERR_STATE, "not an RFF-weight trajectory");
  if (int rc = set_device(h)) return rc;
  HIPCHK(h, hipMemcpy(theta, t->d_theta.p, (size_t)t->F * t->B * sizeof(double), hipMemcpyDeviceToHost));
  return TGP_OK;
}

int tgp_traj_destroy(tgp_traj t) {
  if (!t) return TGP_OK;
  // a garbage collector may destroy the model handle before its trajectories: `t->h` is not touched here
  (void)hipSetDevice(t->device);
  t->d_W.release();
  t->d_b.release();
  t->d_ws.release();
  t->d_v.release();
  t->d_theta.release();
  delete t;
  (void)hipGetLastError();
  return TGP_OK;
}

int tgp_traj_get_v(tgp_traj t, double* v) {
  if (!t || !v) return TGP_ERR_ARG;
  tgp_handle h = t->h;
  if (!t->canonical) return fail(h, TGP_ERR_STATE, "an RFF-weight trajectory has no canonical weights");
  if (int rc = set_device(h)) return rc;
  HIPCHK(h, hipMemcpy(v, t->d_v.p, (size_t)h->N * t->B * sizeof(double), hipMemcpyDeviceToHost));
  return TGP_OK;
}

int tgp_traj_eval(tgp_traj t, const double* Xq, int64_t M, int per_traj_inputs, double* out, int where) {
  if (!t) return TGP_ERR_ARG;
  tgp_handle h = t->h;
  if (M < 0 || (M > 0 && (!Xq || !out))) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (M == 0) return TGP_OK;
  if (!per_traj_inputs && t->B > 16)
    return fail(h, TGP_ERR_SHAPE, "shared-input evaluation supports B <= 16 trajectories, got %d", t->B);
  if (int rc = set_device(h)) return rc;
  const size_t nin = (size_t)M * (per_traj_inputs ? t->B : 1) * h->d;
  const double* dXq;
  double* dout;
  if (int rc = stage_in(h, h->s_in, Xq, nin, where, &dXq)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, out, (size_t)M * t->B, where, &dout)) return rc;
  (void)hipEventRecord(h->ev0, h->stream);
  launch_traj_eval(h->stream, traj_dev(t), dXq, M, per_traj_inputs, dout, nullptr, nullptr, 0);
  (void)hipEventRecord(h->ev1, h->stream);
  h->last_launches = 1;
  h->last_ms = -1.0;
  if (int rc = stage_out_finish(h, dout, out, (size_t)M * t->B, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_traj_value_grad(tgp_traj t, const double* Xq, int64_t P, double* val, double* grad, int where) {
  if (!t) return TGP_ERR_ARG;
  tgp_handle h = t->h;
  if (P < 0 || (P > 0 && (!Xq || !val || !grad))) return fail(h, TGP_ERR_ARG, "bad arguments");
  if (P == 0) return TGP_OK;
  if (int rc = set_device(h)) return rc;
  const int64_t nitems = P * t->B;
  const double* dXq;
  double *dval, *dgrad;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)nitems * h->d, where, &dXq)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out1, val, (size_t)nitems, where, &dval)) return rc;
  if (int rc = stage_out_prepare(h, h->s_out2, grad, (size_t)nitems * h->d, where, &dgrad)) return rc;
  launch_traj_grad(h->stream, traj_dev(t), dXq, nitems, dval, dgrad);
  if (int rc = stage_out_finish(h, dval, val, (size_t)nitems, where)) return rc;
  if (int rc = stage_out_finish(h, dgrad, grad, (size_t)nitems * h->d, where)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

// arg-min of the B trajectories over device-resident candidates into DEVICE slots fv [B], fi [B]: enqueue only
static int enqueue_traj_argmin(tgp_traj t, const double* dXq, int64_t M, int64_t index_base, double* fv, int64_t* fi) {
  tgp_handle h = t->h;
  const int64_t grid = traj_grid(M);
  const int B = t->B;
  HIPCHK(h, h->s_blkv.reserve((size_t)grid * B * sizeof(double)));
  HIPCHK(h, h->s_blki.reserve((size_t)grid * B * sizeof(int64_t)));
  (void)hipEventRecord(h->ev0, h->stream);
  launch_traj_eval(h->stream, traj_dev(t), dXq, M, 0, nullptr, h->s_blkv.as<double>(),
                   h->s_blki.as<int64_t>(), index_base);
  (void)hipEventRecord(h->ev1, h->stream);
  h->last_launches = 1;
  h->last_ms = -1.0;
  launch_argmin_final_multi(h->stream, h->s_blkv.as<double>(), h->s_blki.as<int64_t>(), grid, B, fv, fi);
  return TGP_OK;
}

int tgp_traj_argmin(tgp_traj t, const double* Xq, int64_t M, int64_t index_base, double* best_val,
                    int64_t* best_idx, int where) {
  if (!t) return TGP_ERR_ARG;
  tgp_handle h = t->h;
  if (M < 1 || !Xq) return fail(h, TGP_ERR_SHAPE, "arg-min needs M >= 1 candidates");
  if (t->B > 16) return fail(h, TGP_ERR_SHAPE, "supports B <= 16 trajectories per call, got %d", t->B);
  if (int rc = set_device(h)) return rc;
  const double* dXq;
  if (int rc = stage_in(h, h->s_in, Xq, (size_t)M * h->d, where, &dXq)) return rc;
  const int B = t->B;
  HIPCHK(h, h->s_small.reserve(64 + 2 * 16 * 8));
  double* fv = h->s_small.as<double>();
  int64_t* fi = (int64_t*)(fv + 16);
  if (int rc = enqueue_traj_argmin(t, dXq, M, index_base, fv, fi)) return rc;
  if (best_val) HIPCHK(h, hipMemcpyAsync(best_val, fv, B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (best_idx) HIPCHK(h, hipMemcpyAsync(best_idx, fi, B * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_traj_argmin_async(tgp_traj t, const double* Xq_device, int64_t M, int64_t index_base, double* pairs_device) {
  if (!t) return TGP_ERR_ARG;
  tgp_handle h = t->h;
  if (M < 1 || !Xq_device) return fail(h, TGP_ERR_SHAPE, "arg-min needs M >= 1 candidates");
  if (!pairs_device) return fail(h, TGP_ERR_ARG, "pairs_device is NULL");
  if (t->B > 16) return fail(h, TGP_ERR_SHAPE, "supports B <= 16 trajectories per call, got %d", t->B);
  if (int rc = set_device(h)) return rc;
  if (int rc = enqueue_traj_argmin(t, Xq_device, M, index_base, pairs_device, (int64_t*)(pairs_device + t->B)))
    return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_stream_synchronize(tgp_handle h) {
  if (!h) return TGP_ERR_ARG;
  if (int rc = set_device(h)) return rc;
  if (int rc = sync(h)) return rc;
  HIPCHK(h, hipGetLastError());
  return TGP_OK;
}

int tgp_last_kernel_ms(tgp_handle h, double* ms, int* launches) {
  if (!h) return TGP_ERR_ARG;
  if (h->last_ms < 0.0) {
    (void)hipSetDevice(h->device);  // the events belong to the handle's device
    float f = 0.f;
    if (hipEventElapsedTime(&f, h->ev0, h->ev1) == hipSuccess) h->last_ms = f;
    else h->last_ms = 0.0;
  }
  if (ms) *ms = h->last_ms;
  if (launches) *launches = h->last_launches;
  return TGP_OK;
}

}  // extern "C"
