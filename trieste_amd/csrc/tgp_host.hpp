// Host-side state behind the opaque handles of include/tgp.h, shared by the translation units that
// implement the C-ABI (tgp_api.hip: single-device entry points; tgp_group.hip: the multi-device group).
#pragma once
#include "../../include/tgp.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "tgp_internal.hpp"

namespace tgp {

struct DevBuf {  // grow-only device buffer
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) {
      cap = bytes;
      // TGP_POISON=1 (tests): fill fresh allocations with NaNs so that any read of memory the engine did
      // not write shows up deterministically instead of depending on what the allocator recycled
      static const bool poison = getenv("TGP_POISON") != nullptr;
      if (poison) e = hipMemset(p, 0xFF, bytes);
    }
    return e;
  }
  hipError_t grow_keep(size_t bytes, size_t keep, hipStream_t st) {  // like reserve, but keeps the first `keep` bytes
    if (bytes <= cap) return hipSuccess;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) {
        (void)hipFree(q);
        return e;
      }
    }
    if (p) (void)hipFree(p);
    p = q;
    cap = bytes;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return (T*)p; }
};

}  // namespace tgp

using tgp::DevBuf;

struct tgp_handle_s {
  int device = 0, d = 0, dp = 0, kind = 0, num_cu = 256;
  hipStream_t stream = nullptr;      // where the kernels go (default stream, a caller's, or own_stream)
  hipStream_t own_stream = nullptr;  // created by tgp_use_private_stream, destroyed with the handle
  std::string err;
  // hyper-parameters
  bool have_hyper = false, have_data = false;
  bool have_xy = false;  // X, Y are on the device (tgp_set_data has run): tgp_nlml_trial may re-factorise them
  double variance = 1.0, noise = 1.0, mean_const = 0.0;
  std::vector<double> ls;  // [d]
  int64_t N = 0, Npad = 0;
  // The factorisation only ever writes zeros above the diagonals of d_L / d_W and rewrites every block below them:
  // the buffers are wiped once per (allocation, Npad), not on every update (2 x Npad^2 x 8 bytes of HBM writes)
  const void *zeroed_L = nullptr, *zeroed_W = nullptr;
  int64_t zeroed_npad = 0;
  int variant = 0;
  // arithmetic of the plain (non-joint) sweeps: TGP_PREC_F64, or TGP_PREC_I8X4 = W K* on the int8 matrix cores with
  // four digit planes per operand (tgp_set_precision); the planes of W are rebuilt lazily per factorisation
  int precision = 0;      // the arithmetic in effect (never TGP_PREC_AUTO)
  int precision_req = 0;  // what tgp_set_precision asked for; TGP_PREC_AUTO is resolved per factorisation
  // TGP_PREC_AUTO: the split-precision sweep with the a-posteriori repair, on a ladder 0: four planes, 1: five planes,
  // 2: float64 (tgp_api.hip resolve_precision); a rung is left when a sweep had to recompute too many candidates
  bool repair = false;        // the sweeps of the arithmetic in effect run with the repair
  int auto_level = 0;
  uint64_t auto_epoch = 1;    // bumps whenever the rung changes or the ladder restarts: stale reports are ignored
  bool auto_pinned = false;   // resolved by sweep_blocks for the call in progress
  int64_t* rep_host = nullptr;  // pinned, RS_WORDS words: the stats block (tgp_internal.hpp RS_*) of the last completed repaired sweep
  int64_t rep_last_M = 0, rep_last_count = 0;
  // the canary (tgp_api.hip launch_sweep_i8_timed): of every AUTO sweep a UNIFORM sample (one candidate in 4096) and an
  // ADVERSARIAL one (per 1 / 64 of the sweep the unflagged candidate whose bound sits closest to its tolerance) are recomputed
  // in float64 and compared with the bounds the int8 kernel priced them at; a violation in either demotes the ladder.  The
  // samples are compared, never scattered, and the uniform offset is a function of (N, hyper-parameters, M, rung): a sweep's outputs
  // and the ladder's decisions are a pure function of (model state, inputs).
  double auto_sigma = 8.0;       // K_SIGMA of the per-candidate bound (tgp_set_auto_sigma)
  uint64_t canary_epoch = 0;     // the rung (auto_epoch) whose canary words are live on the device
  // [0] the uniform stratum, [1] the adversarial one
  int64_t can_checked[2] = {0, 0}, can_viol[2] = {0, 0};              // the current rung, as of the last report read
  int64_t can_checked_total[2] = {0, 0}, can_viol_total[2] = {0, 0};  // earlier rungs / epochs since tgp_set_precision / tgp_set_auto_sigma
  double can_worst[2] = {0.0, 0.0};   // worst |d var| / (bound + slack) seen
  int64_t can_slack = 0, can_slack_total = 0;   // samples inside bound + slack but outside the bound alone (float64 rounding)
  int can_demotions = 0;         // rungs left BECAUSE of a violation
  std::vector<double> auto_hyp;  // (variance / noise, lengthscales) when a rung was last left: the rung is kept while the
                                 // hyper-parameters in effect at the next SWEEP stay within a factor two of these
  bool auto_hyper_dirty = false; // tgp_set_hyper ran since the last sweep: the keep-or-restart decision is taken at the next
                                 // sweep (a fit's trial evaluations move the hyper-parameters far and back again)
  DevBuf s_rep, s_rep_stats, s_blkctr;   // (s_blkctr: the int8 sweep's candidate-block counter, one word)
  DevBuf d_wq, d_rs, d_xsa;
  uint64_t wq_version = 0;
  int wq_planes = 0;
  // model state on device
  DevBuf d_xn, d_ls, d_X, d_Y, d_Xs, d_A, d_L, d_W, d_alpha, d_err, d_tmp1, d_tmp2, d_info;
  // local penalization applied to every tgp_acq_* result while pen_kind != 0 (tgp_set_penalization)
  int pen_kind = 0, pen_P = 0;
  DevBuf d_pen;  // [P, d] pending points, [P] radius, [P] scale
  // entropy-search tails (TGP_ACQ_MES / TGP_ACQ_GIBBON): min-value samples; GIBBON's repulsion twin
  int ent_S = 0;
  DevBuf d_ent;  // [S]
  tgp_handle rep_twin = nullptr;  // not owned: this model conditioned additionally on the pending points
  double rep_weight = 0.0;
  // when the twin is literally this model's data + m <= 16 appended rows (same hyper-parameters), its variance is
  // a rank-m correction of this model's: checked once per (data, twin data) version pair
  uint64_t data_version = 0;       // process-wide unique stamp of the current factorisation (0: none)
  uint64_t rep_self_version = 0, rep_twin_version = 0;
  bool rep_checked = false, rep_lowrank = false;
  int rep_m = 0;
  DevBuf d_repv;                   // [N + m][m]: the twin's last m rows of W as weight columns
  // scratch
  DevBuf s_ent, s_in, s_in2, s_out1, s_out2, s_out3, s_blkv, s_blki, s_small, s_kcache, s_aslab, s_grad, s_ks, s_part;
  // `update` as one persistent launch: the task list of the current block count (tgp_kernels_dag.hip)
  // one cached plan per use: 0 the full update, 1 the factor-only trial (tgp_nlml_trial), 2 .. 9 the batched
  // factor-only launch (tgp_nlml_trial_batch) per member count -- a fit alternates between them (90 draws are eleven
  // launches of eight and one of two), and building a plan costs as much as the update
  struct DagPlan {         // (a view: the device arrays belong to the process-wide plan cache of tgp_api.hip -- plans are
    int nb = 0, ntasks = 0, grid = 0, B = 0;   //  immutable once built, every handle of a device shares them)
    bool split = false;                         // the round-6 split plan (tgp_api.hip dag_split)
    bool duo = false;                           // ... launched with the two-workgroup chain (dag_duo)
    int64_t ld = 0;
    const void *tasks = nullptr, *chain = nullptr, *topo = nullptr;
  } dag_plan[2 + 48];  // 0 full, 1 factor-only, 2 + (B - 1): batched factor-only with B = 1 .. TRIAL_BATCH_MAX members
  int dag_last_slot = 0;  // the slot of the most recent launch (its error words are read back after the stream drains)
  int update_share = 1;  // tgp_set_update_concurrency: the persistent update kernel takes num_cu / update_share workgroups
  DevBuf d_dag_flags, d_dag_trace;
  size_t dag_state_words = 0;  // d_dag_flags: B x [ntasks + 2 NB] flag words, control words, start counts (DagArgs), zeroed per launch
  // batched trial evaluations: [B][3] matrices (A, L, W), [B] scaled inputs / centred targets / z, per-member scalars
  // (they live in a PROCESS-WIDE scratch per device, tgp_api.hip BatchScratch: every fit of every model reuses it)
  // timing of the dominant kernel
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0.0;
  int last_launches = 0;
};

struct tgp_traj_s {
  tgp_handle h = nullptr;
  int F = 0, B = 0;
  DevBuf d_W, d_b, d_ws, d_v, d_theta;
  int canonical = 1;
  int device = 0;  // copied from the handle: destruction must not dereference `h` (it may be gone already)
};

// ---- internal entry points shared between the translation units (not part of the C-ABI) -------------
namespace tgp {
// set the thread's device to the handle's and clear the thread's sticky HIP error; TGP_OK or TGP_ERR_HIP
int host_set_device(tgp_handle h);
// record `msg` as the handle's last error and return `code`
int host_fail(tgp_handle h, int code, const char* fmt, ...);
}  // namespace tgp
