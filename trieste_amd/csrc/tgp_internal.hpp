// Internal launch interface between tgp_api.hip (host side of the C-ABI) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace tgp {

constexpr int SW_BN = 128;  // candidates per workgroup (sweep kernel)
constexpr int NPAD_MULT = 256;  // row block of W per workgroup tile of the sweep
constexpr int LEAF = 64;    // Cholesky / inverse leaf block
constexpr int MAX_D = 32;
constexpr int MAX_Q = 64;

inline int dpad_of(int d) {
  if (d <= 2) return 2;
  if (d <= 4) return 4;
  if (d <= 6) return 6;
  if (d <= 8) return 8;
  if (d <= 16) return 16;
  return 32;
}

struct ModelDev {       // device-resident model state (all pointers device)
  int kind, d, dp;      // kernel kind, input dim, padded dim
  int64_t N, Npad;
  double variance, noise, mean_const;
  const double* ls;     // [dp] lengthscales padded with 1.0
  const double* Xs;     // [Npad][dp]  X / ls, zero padded
  const double* xn;     // [Npad] |Xs_k|^2 (dot-product form of the distances in the trajectory kernel)
  const double* Wt;     // [Npad][Npad] Wt[k][i] = (L^-1)[i][k], zero outside the N x N lower part
  const double* alpha;  // [Npad] K^-1 (Y - c), zero padded
};

struct SweepArgs {
  ModelDev m;
  const double* Xq;   // [M][d] raw candidates (device)
  int64_t M;
  double* mean_out;   // [M] or null
  double* var_out;    // [M] or null
  double* acq_out;    // [M] or null
  int acq_kind;       // -1: none
  double acq_param;
  double* blk_val;    // [grid] or null : per-workgroup best acquisition value
  int64_t* blk_idx;   // [grid]
  int64_t index_base;
  // joint mode
  int q;              // group size (joint mode only)
  int64_t G;          // number of groups
  double* cov_out;    // [G][q][q]
  double* kcache;     // [grid][Npad][128] per-workgroup K* slabs (device scratch; int8 sweep: [grid][Npad][64][4] bytes)
  const void* i8_wq;  // int8 sweep: digit planes of W, Wq[s][k/32][i][32] (4 or 5 planes of Npad^2 bytes)
  const double* i8_rs; // int8 sweep: [Npad] row scales S_i = 2 max_k |W_ik|
  const double* i8_xsa; // int8 sweep: [Npad / 32] tiles (32 rows of Xs + alpha) for LDS staging, or null (scalar loads)
  unsigned* blk_ctr;    // sweep_i8_kernel: a workgroup's candidate blocks after its first are drawn from this counter (zero at
                        // launch); null: blocks i, i + #WG, ...  (sweep_dma_kernel, 8 ms per block: measured neutral, not wired)
  double* aslab;      // [grid][Npad][128] C = W K* slabs of joint mode (device scratch)
  // row-group split of small sweeps (SPLIT instantiation): group g of a candidate block owns the row
  // blocks [split_ib[g], split_ib[g+1]) of W and leaves partial (mean, sum c^2) in `part`
  int split_g;            // 0/1: off
  int split_ib[17];       // up to 16 groups (8 for the ordinary small sweeps; 16 for the repair pass of TGP_PREC_AUTO)
  double* part;           // [blocks][split_g][2][128]
  // a-posteriori repair of the split-precision sweep (TGP_PREC_AUTO, tgp_api.hip sweep_i8_repaired): the int8
  // kernel prices every candidate's own truncation error on the variance,
  //     e_var = rep_scale sqrt(sum_i c_i^2 S_i^2 (i + 1)),  rep_scale = K_SIGMA 2 2^-32.8 S' (four planes),
  // and leaves rep_ub[j] = +inf where e_var exceeds 1e-5 var + rep_floor (the candidate is recomputed in float64),
  // else the upper end of its acquisition value's interval; the block winners then carry the LOWER ends.
  double* rep_ub;         // [M] or null (null: the plain emulated-precision sweep, no bounds)
  double rep_scale, rep_floor;
  // the canary of TGP_PREC_AUTO: candidate j is a sample when (j + canary_off) % 4096 == 0; the int8 kernel leaves its
  // (variance, bound) in canary_rec[2 ((j + canary_off) / 4096)], it is recomputed in float64 with the repair list and
  // repair_canary_kernel compares the two (null: no samples)
  double* canary_rec;
  int64_t canary_off;
  // the canary's adversarial stratum (round 6): per 64-candidate block of the int8 sweep (ratio = bound / tolerance, index as
  // the bits of an int64, int8 variance, bound) of its worst UNFLAGGED candidate (ratio < 0: none); null: no stratum
  double* adv_rec;
  const int64_t* M_dev;   // SPLIT instantiation + combine kernel: the candidate count lives on the device (the
                          // repair pass over the flagged candidates is enqueued without a host round trip)
};
constexpr double I8_TIGHT = 1.0078125;  // digit-plane scales S_i = I8_TIGHT max_k |W_ik|, S' = I8_TIGHT variance: the
                                        // balanced digits reach |q| <= 0x7f7f7f7f = 0.99609 2^31 > 2^31 / I8_TIGHT
constexpr int64_t I8_CANARY_PERIOD = 4096;  // one sampled candidate in 4096 (a power of two)
void launch_sweep_combine(hipStream_t s, const SweepArgs& a, int64_t nblk);

// ---- linalg (tgp_kernels_linalg.hip) ----
void launch_scale_inputs(hipStream_t s, const double* X, const double* ls, double* Xs, int64_t N,
                         int64_t Npad, int d, int dp);
void launch_assemble_K(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp,
                       int kind, double variance, double noise, int64_t row0 = 0);
void launch_assemble_K_batch(hipStream_t s, const double* Xs, double* A, int64_t N, int64_t Npad, int dp, int kind,
                             const double* hyp, int64_t hyp_stride, int64_t a_stride, int B);
void launch_batch_prep(hipStream_t s, const double* X, const double* Y, const double* hyp, int64_t hyp_stride, int B,
                       double* Xs, double* err, int64_t N, int64_t Npad, int d, int dp);
void launch_leaf(hipStream_t s, const double* A, double* L, double* W, int64_t ld, int64_t off,
                 int* info);
void launch_leaf128(hipStream_t s, const double* A, double* L, double* W, int64_t ld, int64_t off, int* info);
// C[m x n] = alpha * A[m x k] * op(B) + beta * C ;  TB: B stored [n x k] (row-major), else [k x n]
void launch_gemm(hipStream_t s, bool tb, int m, int n, int k, double alpha, const double* A, int64_t lda,
                 const double* B, int64_t ldb, double beta, double* C, int64_t ldc, bool lower_only, int tri = 0,
                 int small_tiles = 0);
void launch_node_pair(hipStream_t s, int s2, int s1, const double* L21, double* A22, const double* W11, double* T,
                      int64_t ld);
void launch_gemm_ksplit(hipStream_t s, bool tb, int m, int n, int k, double alpha, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int tri, int nz,
                        double* scratch);
void launch_transpose_mask(hipStream_t s, const double* W, double* Wt, int64_t N, int64_t Npad);
void launch_zero(hipStream_t s, double* p, int64_t n);
void launch_center(hipStream_t s, const double* Y, double c, double* err, int64_t N, int64_t Npad);
void launch_transpose(hipStream_t s, const double* src, int64_t rows, int64_t cols, int64_t lds, double* dst,
                      int64_t ldd);
void launch_cov_tail(hipStream_t s, const ModelDev& m, const double* X1, int64_t P1, const double* X2,
                     int64_t P2, const double* S, int64_t lds, double* out);
void launch_cov_sym_tail(hipStream_t s, const ModelDev& m, const double* X, int64_t n, int64_t Pp, const double* S,
                         double jitter, double* A);
void launch_pad_copy(hipStream_t s, const double* src, int64_t r, int64_t c, double* dst, int64_t rp, int64_t cp);
void launch_sample_tail(hipStream_t s, const double* mean, const double* R, int64_t n, int S, int64_t Sp, double* out);
void launch_row_norms(hipStream_t s, const double* Xs, double* xn, int64_t Npad, int dp);
// y[i] = sum_k M[i][k] x[k] over k in [klo(i), khi(i)] ; lower: k<=i ; upper: k>=i
void launch_trmv(hipStream_t s, const double* Mx, int64_t ld, int64_t n, const double* x, double* y,
                 bool lower);
void launch_axpby_vec(hipStream_t s, int64_t n, double a, const double* x, double b, const double* y,
                      double* out);

// ---- sweep (tgp_kernels_sweep_*.hip) ----
int64_t sweep_grid(const SweepArgs& a, bool joint);

// ---- misc (tgp_kernels_misc.hip) ----
void launch_predict_mean(hipStream_t s, const ModelDev& m, const double* Xq, int64_t M, double* mean);
void launch_argmax_final(hipStream_t s, const double* blk_val, const int64_t* blk_idx, int64_t n,
                         double* out_val, int64_t* out_idx);
void launch_penalize(hipStream_t s, double* vals, const double* Xq, int64_t M, int d, int kind, int P,
                     const double* pend, const double* radius, const double* scale, bool init_one = false);
void launch_penalize_grad(hipStream_t s, double* val, double* grad, const double* Xq, int64_t Pq, int d, int kind,
                          int P, const double* pend, const double* radius, const double* scale);
void launch_min_value(hipStream_t s, const double* v, int64_t n, double* out);
void launch_topk_pass(hipStream_t s, const double* vals, int64_t M, int64_t index_base, const double* prev_val,
                      const int64_t* prev_idx, double* scratch_val, int64_t* scratch_idx, double* out_val,
                      int64_t* out_idx);
void launch_topk_small(hipStream_t s, const double* vals, int64_t M, int64_t index_base, int k, double* out_val,
                       int64_t* out_idx, double* scratch_val, int64_t* scratch_idx);
int64_t topk_small_max();
void launch_sample_box(hipStream_t s, uint64_t seed, int64_t first, int64_t M, int d,
                       const double* lower, const double* upper, double* out);
void launch_qei_tail(hipStream_t s, const double* mean, const double* cov, int64_t G, int q,
                     const double* eps, int S, double eta, double jitter, double* out, double* samples_out,
                     int* info);
size_t qei_grad_tail_lds_bytes(int q, int S);
void launch_qei_grad_tail(hipStream_t s, const double* mean, const double* cov, int64_t G, int q, const double* eps, int S,
                          double eta, double jitter, double* val, double* gmean, double* gcov, int* info);
// gradients (tgp_kernels_grad.hip)
void launch_kstar_t(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, double* B);
size_t predict_small_scratch_doubles(int64_t Ppad);   // `part` of launch_predict_small_tail
void launch_predict_small_tail(hipStream_t s, const ModelDev& m, int64_t P, int64_t Ppad, const double* B, const double* C1,
                               double* part, double* mean_out, double* var_out);
void launch_joint_pick(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, int q, const double* S, double* cov);
void launch_joint_mix(hipStream_t s, const double* C1, const double* gcov, int64_t P, int64_t Ppad, int64_t Npad, int q, double* D);
void launch_joint_vjp_tail(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad, int q, const double* B,
                           const double* C1, const double* Z, double* part, const double* gmean, const double* gcov, double* grad);
size_t grad_tail_scratch_doubles(int64_t Ppad);   // `part` of launch_grad_tail
void launch_grad_tail(hipStream_t s, const ModelDev& m, const double* Xq, int64_t P, int64_t Ppad,
                      const double* B, const double* C1, const double* Z, double* part, int acq, double param, double* val,
                      double* grad, const double* samples = nullptr, int S = 0, double rep_w = 0.0,
                      double accum = 0.0);
void launch_entropy_tail(hipStream_t s, const double* mean, const double* var, const double* var_twin, int64_t M,
                         int acq, double noise, const double* samples, int S, double weight, double* out);
int64_t nlml_blocks(int64_t Npad);
void launch_nlml(hipStream_t s, const ModelDev& m, const double* Kinv, const double* L, const double* err,
                 double* partial, double* out);
void launch_nlml_value(hipStream_t s, const ModelDev& m, const double* L, const double* err, double* out);
void launch_nlml_value_batch(hipStream_t s, const ModelDev& m, const double* L, const double* z, double* out, int B,
                             int64_t l_stride, int64_t v_stride, int64_t o_stride);
// trajectories
struct TrajDev {
  ModelDev m;
  int F, B;
  const double* rffW;  // [F][dp] zero padded
  const double* rffb;  // [F]
  const double *rffW_ht, *rffb_ht;  // the same basis in half turns: W / pi, b / pi + 1/2 (evaluation kernel)
  const double* ws;    // [F][B]  sqrt(2 variance / F) * w
  const double* v;     // [Npad][B] canonical weights, zero padded
  int canonical;       // 1: decoupled trajectory (features + k(x, X) v); 0: RFF-only trajectory (features + mean)
};
void launch_rff_project(hipStream_t s, const TrajDev& t, const double* Xs_pts, int64_t npts,
                        double* out /*[npts][B]*/);
void launch_traj_eval(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, int per_traj,
                      double* out, double* blk_val, int64_t* blk_idx, int64_t index_base);
void launch_kernel_sums(hipStream_t s, const TrajDev& t, const double* Xq, int64_t M, double* out);
void launch_lowrank_var(hipStream_t s, const double* var, const double* u, int64_t M, int m, double* var_out);
void launch_rows_to_columns(hipStream_t s, const double* W, int64_t ld, int64_t row0, int m, int64_t n, double* out);
void launch_prefix_differs(hipStream_t s, const double* a, const double* b, int64_t n, int* flag);
int64_t traj_grid(int64_t M);
void launch_rff_features(hipStream_t s, const TrajDev& t, double scale, int64_t Fp, double* Phi);
void launch_sym_finish(hipStream_t s, double* A, int64_t n, int64_t np, double scale, double shift, int negate);
void launch_theta_tail(hipStream_t s, const double* mean, int64_t ldm, const double* R, int64_t ldr, int F, int B,
                       double scale, double* theta, double* ws);
void launch_traj_grad(hipStream_t s, const TrajDev& t, const double* Xq, int64_t nitems, double* val,
                      double* grad);
// ---- `update` as one persistent launch (tgp_kernels_dag.hip) -----------------------------------------------------
constexpr int DAG_MAT_A = 0, DAG_MAT_L = 1, DAG_MAT_W = 2;   // which matrix a tile offset refers to
constexpr uint32_t DAG_NN = 1, DAG_BETA = 2, DAG_NEG = 4;     // B operand natural (else transposed); add Cin; negate
constexpr uint32_t DAG_HALF = 8, DAG_HI = 16;                 // round 6: the task computes 64 of the tile's 128 rows (DAG_HI: rows 64 ..)
constexpr uint32_t DAG_SIB = 32;   // (an upper half) dep3 is NOT a start dependency: it is the flag of the task's lower-half sibling, which the
                                   // task waits for AT ITS END, before its own flag goes up -- that flag then says "both halves are done"
                                   // and whoever needs the whole tile waits for this one flag (the sibling is earlier in the list)
constexpr int DAG_CTRL_WORDS = 64;
constexpr int DAG_DUO_PF = 32;   // the two-workgroup chain: panel flag words per block row ([8 panels][panel wave 0, panel wave 1, W_d's wave, -])
struct DagTask {  // one 128 x 128 tile task: out = beta Cin + alpha sum_{kt < nk} A_kt B_kt(^T); 48 bytes
  uint32_t a_off, b_off, c_off, o_off;  // element offsets of the first tiles (k tiles of A follow at +128; of B at
                                        // +128 (transposed form) or +128 ld (natural form))
  uint32_t nk, flags;
  uint8_t a_mat, b_mat, c_mat, o_mat;
  uint32_t dep[3];                      // flags to wait for (0xffffffff: none)
  uint32_t set;                         // flag to raise when the tile is stored
  uint32_t dep3;                        // a fourth flag to wait for (round 6: consumers of a tile produced in two halves)
};
struct DagArgs {
  double *Ap, *Lp, *Wp;        // K + s I (in; its tiles carry the partial sums P), L, W = L^-1; all ld x ld, row-major
  int64_t ld;
  int NB, ntasks;              // 128-blocks per side; tile tasks
  const DagTask* tasks;        // [ntasks]
  const uint32_t* chain_dep;   // [3 NB] flag the chain workgroup waits for before step j's leaf [2 j] / its L(j+1,j) [2 j + 1];
                               // [2 NB + j]: a second flag before L(j+1,j) (P(j+1,j) finished in two halves), or none
  const uint32_t* topo;        // [ntasks] dispatch order: task indices in a topological order (dag_build)
  uint32_t* flags;             // [ntasks + 2 NB], zero at launch
  uint32_t* ctrl;              // [DAG_CTRL_WORDS] control words (arrival ticket, error, list head), then [ntasks]
                               // start counts (then [DAG_DUO_PF NB] panel flags when `duo`); ALL ZERO at launch (memset before every launch)
  int* info;                   // Cholesky breakdown report (1 + index of the first bad pivot)
  unsigned long long* trace;   // development aid (TGP_DAG_TRACE): [NB][32] chain + [ntasks][4] task time stamps, or null
  // batched launch (tgp_nlml_trial_batch): B > 1 members share the plan; member b's matrices are Ap / Lp / Wp +
  // b mat_stride, its flags `flags + b flags_stride`, its breakdown report info[b]; topo then holds B ntasks entries
  // (member << 24 | task) and the start counts are [B][ntasks]
  int B;
  int duo;                     // round 6: the chain is TWO workgroups (tgp_kernels_dag.hip run_duo; B <= 1 only); DAG_DUO_PF NB panel flag
                               // words then follow the start counts (zero at launch like everything else)
  int64_t mat_stride;
  uint32_t flags_stride;
};
void dag_merge_order(const std::vector<uint32_t>& member_order, int B, std::vector<uint32_t>& merged);
void dag_build(int NB, int64_t ld, std::vector<DagTask>& tasks, std::vector<uint32_t>& chain_dep, int& n_urgent,
               std::vector<uint32_t>* topo_out = nullptr, int workers = 255, bool with_inverse = true, int batch = 1,
               int batch_workers = 0, std::vector<uint32_t>* batch_out = nullptr, bool split_critical = false, bool duo = false);
// z = L^-1 r over 128-blocks from L and the diagonal inverses in W; flags: [NB] words, zero at launch
void launch_block_trsv(hipStream_t s, const double* L, const double* W, int64_t ld, int NB, const double* r, double* z,
                       uint32_t* flags, int B = 1, int64_t mat_stride = 0);
hipError_t launch_dag_update(hipStream_t s, const DagArgs& a, int grid);
size_t dag_lds_bytes();

// rs: [2][Npad] -- row scales S_i, then the row weights S_i^2 (i + 1) of the a-posteriori error model
void launch_w_digits(hipStream_t s, const double* W, int64_t N, int64_t Npad, double* rs, void* Wq, int planes);
// [Npad / 32] tiles of xt doubles: 32 rows of Xs, their alpha, their squared norms, zero padding (the int8 sweep stages them
// in LDS by DMA)
void launch_xs_tiles(hipStream_t s, const double* Xs, const double* alpha, int64_t Npad, int dp, int xt, double* out);
constexpr int i8_xs_tile_doubles(int dp) { return ((32 * dp + 64 + 127) / 128) * 128; }
// LDS budget of sweep_i8_kernel (tgp_kernels_sweep_i8.inc), shared with the launch code of tgp_api.hip: three stages of NS
// digit planes of a [256 rows + 64 candidates] x 32 B tile, the scaled candidate coordinates [dp][64], and -- when they fit --
// the staged training-row tiles of the generating steps: three buffers beside four planes, TWO beside five (round 6), plus the
// candidates' squared norms [64], which live in the first tile buffer's padding when that has 64 spare doubles
constexpr int i8_stage_bytes(int ns) { return ns * (256 + 64) * 32; }
constexpr int i8_tile_buffers(int ns) { return ns == 4 ? 3 : 2; }
constexpr bool i8_qn_in_pad(int dp) { return i8_xs_tile_doubles(dp) - (32 * dp + 64) >= 64; }
constexpr int i8_lds_bytes(int ns, int dp, bool tiles) {
  return 3 * i8_stage_bytes(ns) + 64 * dp * 8 +
         (tiles ? i8_tile_buffers(ns) * i8_xs_tile_doubles(dp) * 8 + (i8_qn_in_pad(dp) ? 0 : 512) : 0);
}
constexpr bool i8_tiles_fit(int ns, int dp) { return dp <= 16 && i8_lds_bytes(ns, dp, true) <= 160 * 1024; }
// ---- a-posteriori repair of the split-precision sweep (tgp_kernels_misc.hip) ----
// stats [8]: {count (zeroed here), M, tag, canary violations, canaries checked, worst |d var| / bound as the bits of a
// double, -, -}; the canary words accumulate until `reset_canary`
void launch_repair_begin(hipStream_t s, int64_t* stats, int64_t M, int64_t tag, bool reset_canary);
// stats words (int64 unless said otherwise): the layout of the 16-word block launch_repair_begin owns
constexpr int RS_COUNT = 0, RS_M = 1, RS_TAG = 2, RS_VIOL = 3, RS_CHECKED = 4, RS_WORST = 5 /* bits of a double */,
              RS_ADV_VIOL = 6, RS_ADV_CHECKED = 7, RS_L = 8 /* double L, int64 index */, RS_ROUTE = 10 /* two counts */,
              RS_ADV_WORST = 12 /* bits of a double */, RS_SLACK_SAVED = 13, RS_WORDS = 16;
constexpr int I8_ADV_GROUPS = 64;   // the adversarial stratum: the worst bound / tolerance of every 1 / 64 of the sweep
// adversarial stratum: group g of <= 64 scans its share of the nblk block records (adv_rec of the int8 sweep), appends its
// worst candidate to the repair list (stats[0]) and leaves adv_sel[g] = {list position or -1 (as the bits of an int64), its
// int8 variance, its bound}
void launch_repair_adv(hipStream_t s, const double* adv_rec, int64_t nblk, int64_t* list, int64_t* stats, double* adv_sel);
// list [<= M]: indices j with ub[j] == +inf or ub[j] >= *L (L null: only +inf) or (j + canary_off) % 4096 == 0
// (canary_off < 0: no samples); count = stats[0]
void launch_repair_flag(hipStream_t s, const double* ub, int64_t M, const double* L, int64_t* list, int64_t* stats,
                        int64_t canary_off);
// the sampled candidates among list[0 .. stats[0]): |rvar[r] - rec var| against rec bound + slack -> stats[3..5]
void launch_repair_canary(hipStream_t s, const int64_t* list, int64_t* stats, int64_t cap, const double* rvar,
                          const double* rec, int64_t canary_off, double slack, const double* adv_sel);
// the repair of a few candidates as a product (tgp_kernels_misc.hip): count routing, K*^T with the count on the device,
// column sums + the sweep's tail
void launch_repair_route(hipStream_t s, const int64_t* stats, int64_t pcap, int64_t* route);
void launch_kstar_t_dev(hipStream_t s, const ModelDev& m, const double* Xq, const int64_t* P_dev, int64_t Ppad, double* B);
void launch_repair_product_tail(hipStream_t s, const ModelDev& m, const double* B, const double* C, int64_t Ppad,
                                const int64_t* P_dev, int acq_kind, double acq_param, double* part, double* mean_out,
                                double* var_out, double* acq_out);
int64_t repair_product_part_doubles(int64_t Ppad);
void launch_repair_gather(hipStream_t s, const double* Xq, int d, const int64_t* list, const int64_t* count, int64_t cap,
                          double* Xg);
// only what the sweep FLAGGED is written over the int8 results (ub[j] == +inf, or ub[j] >= *L in a fused arg-max): the
// canary's samples are compared, not scattered, so that a sweep's outputs do not depend on which candidates it sampled
void launch_repair_scatter(hipStream_t s, const int64_t* list, const int64_t* count, int64_t cap, const double* rmean,
                           const double* rvar, const double* racq, double* mean, double* var, double* acq, const double* ub,
                           const double* L);
// per-slot (max value, min index) partials of vals [M] (NaN never wins) into blk_val / blk_idx [nslots]
void launch_values_argmax(hipStream_t s, const double* vals, int64_t M, int64_t index_base, double* blk_val,
                          int64_t* blk_idx, int64_t nslots);
void launch_w_absmax(hipStream_t s, const double* W, int64_t N, int64_t Npad, double* out);  // *out zeroed before
void launch_merge_winners(hipStream_t s, const double* gathered, int P, int V, int minimize, double* out);
void launch_argmin_final_multi(hipStream_t s, const double* blk_val, const int64_t* blk_idx,
                               int64_t nblk, int B, double* out_val, int64_t* out_idx);

}  // namespace tgp
