// `update` as ONE persistent launch: Cholesky factor L of K + s I AND W = L^-1 as a static DAG of 128 x 128 tile tasks.
//
// Reference call site: gpflow's GPR posterior cache (tf.linalg.cholesky(K + s I)) reached from
// trieste/models/gpflow/models.py:171-186 -> interface.py:108-112; called 10^2..10^3 times per BO step
// through optimize_encoded (models.py:256-321), which is why its latency matters.
//
// The recursion of tgp_api.hip (chol_inv) is a chain of ~130 dependent launches at N = 4096 in which the only true
// chain -- leaf(j) -> L(j+1,j) -> S(j+1,j+1) -> leaf(j+1) -- leaves 255 of 256 CUs idle a third of the time, and every
// other product waits at launch boundaries it does not depend on.  Here:
//   * one workgroup per CU, all resident; the first to arrive is the CHAIN workgroup, the others are BULK workers;
//   * the chain workgroup walks the block diagonal: S = P(j,j) - L(j,j-1) L(j,j-1)^T (in LDS, operands never leave the
//     CU), the 128-leaf (factor + inverse of the diagonal block, tgp_leaf_dev.inc), then L(j+1,j) = P(j+1,j) W_jj^T,
//     which stays in LDS for the next step: no memory hand-off ON the chain, only flags going out;
//   * everything else is a generic tile task  C = beta Cin + alpha sum_k A_k B_k(^T)  (k tiles of 128) executed by the
//     bulk workers from ONE static list in a topological order (tools/dag_sim.py is the design model of it):
//       G  P(i,j) -= sum_{k in burst} L(i,k) L(j,k)^T     left-looking bursts (4 / 8 / 16 products per read-modify-write)
//       T  L(i,j)  = P(i,j) W_jj^T                         i >= j + 2 (the chain does i = j + 1)
//       X  V(i,c) += sum_{k in burst} L(i,k) W(k,c)        the inverse, row by row, V kept in W's own tile
//       E  W(i,c)  = -W_ii V(i,c)
//     a worker pops the next list index with one atomic, waits (relaxed agent-scope polls, bounded) for the <= 3 flags
//     the task names -- always tasks earlier in the list or chain steps, so whatever the residency nothing can
//     deadlock -- runs it and sets the task's own flag;
//   * every tile has ONE writer at a time and every partial sum a fixed order: the result does not depend on the
//     schedule (bit-identical run to run and across handles -- the replica contract of SURVEY 8e).
// Memory model (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility"): the L2s of the eight XCDs are not coherent
// and a CU's L1 is never refreshed, so every tile that crosses workgroups inside the launch is WRITTEN write-through
// (sc1 stores, every storing wave drains vmcnt, barrier, then ONE lane stores the flag) and READ past the L1 (sc1
// loads / sc1 LDS-DMA); flags are relaxed agent-scope atomics, zeroed by a memset node before every launch.
// GEMM tasks: 8 waves, wave tile 64 x 32 (the sweep kernel's), k chunks of 32 staged global -> LDS by
// global_load_lds_dwordx4 (two stages, one barrier per chunk); a DMA image is lane-linear, so the row-major operand
// tiles are stored four rows per KiB with the 16-byte granules of row q rotated by 4 q -- conflict-free ds_read_b64.
#include <algorithm>
#include <queue>
#include <vector>

#include <cstdlib>
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

#ifndef TGP_DAG_STAGGER
#define TGP_DAG_STAGGER 1
#endif
#ifndef TGP_DAG_LATE_K4
#define TGP_DAG_LATE_K4 3   // the k4 step (of 8 per chunk) after which the late waves request the next chunk
#endif
#ifndef TGP_DAG_SETPRIO
#define TGP_DAG_SETPRIO 1   // s_setprio around a chunk's MFMAs: N = 8192 update 7.52 -> 7.42 ms (profiles/r04_dag_stagger_ab.txt)
#endif

namespace tgp {
namespace {

#include "tgp_leaf_dev.inc"

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

constexpr uint32_t NONE = 0xffffffffu;
constexpr int CT = 32;  // trace words per chain step (development aid)
constexpr int TILE = 128, KC = 32;                       // tile side, k chunk
constexpr int GROUP_B = 1024 + 16;                       // four operand rows of 32 doubles + 16 B pad
constexpr int STAGE_A = (TILE / 4) * GROUP_B;            // 33 280 B: A tile chunk [128 rows][32 k]
constexpr int BN_ROW = 1024 + 128;                       // NN B operand: one k row of 128 doubles + 128 B pad
constexpr int STAGE_B = KC * BN_ROW;                     // 36 864 B (>= the NT layout's 33 280)
constexpr int STAGE = STAGE_A + STAGE_B;                 // 70 144 B
constexpr int OUT_LD = 130;                              // epilogue tile [128][130] doubles = 133 120 B
constexpr int LEAF_BYTES = QN * QS * 8 + QB * NWORK * WSLOTS * (int)sizeof(WorkItem);  // 141 312 B
constexpr int CTL_OFF = LEAF_BYTES;                      // four control words behind everything
constexpr int DAG_LDS = LEAF_BYTES + 64;
static_assert(2 * STAGE <= LEAF_BYTES && TILE * OUT_LD * 8 <= LEAF_BYTES, "LDS layout");
constexpr unsigned SPIN_LIMIT = 1u << 22;                // ~1.2 us per poll: several seconds, then an error

// error codes in ctrl[2]
constexpr uint32_t DAG_ERR_TIMEOUT = 1;

typedef __attribute__((address_space(1))) double gdouble;  // global memory, said so: no flat instructions
#define DAG_LDS_DECL extern __shared__ __attribute__((aligned(16))) char dag_lds[]
// Values that are the same in every lane but reach a non-inlined function through memory or VGPRs: back to SGPRs
// (buffer resources and the `nn` branch want them there).
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class P>
__device__ __forceinline__ P* uniptr(P* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
  return (P*)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ uint32_t ld_flag(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_flag(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: wait until flags[id] != 0; false on timeout or when another workgroup has raised the error word
__device__ bool wait_flag_at(const uint32_t* flags, uint32_t* ctrl, uint32_t id) {
  if (id == NONE) return true;
  unsigned spins = 0;
  while (ld_flag(flags + id) == 0) {
    __builtin_amdgcn_s_sleep(8);
    if ((++spins & 255u) == 0) {
      if (ld_flag(ctrl + 2) != 0) return false;
      if (spins > SPIN_LIMIT) {
        st_flag(ctrl + 2, DAG_ERR_TIMEOUT);
        st_flag(ctrl + 3, 0x80000000u | id);
        return false;
      }
    }
  }
  return true;
}
__device__ bool wait_flag(const DagArgs& a, uint32_t id) { return wait_flag_at(a.flags, a.ctrl, id); }

// member b of a batched launch (DagArgs::B > 1): its own matrices, flags and breakdown report; the plan (tasks, chain
// dependencies), the control words and the dispatch list are shared
__device__ __forceinline__ DagArgs dag_member(const DagArgs& a, uint32_t b) {
  DagArgs m = a;
  m.Ap = a.Ap + (int64_t)b * a.mat_stride;
  m.Lp = a.Lp + (int64_t)b * a.mat_stride;
  m.Wp = a.Wp + (int64_t)b * a.mat_stride;
  m.flags = a.flags + (size_t)b * a.flags_stride;
  m.info = a.info + b;
  m.trace = nullptr;
  return m;
}

// development aid: 100 MHz wall-clock stamps of the phase boundaries (one lane; only when a trace buffer is given)
__device__ __forceinline__ void stamp(unsigned long long* slot) {
  if (slot) *slot = wall_clock64();
}

__device__ __forceinline__ void glds16_sc1(const void* gsrc, uint32_t lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_uniform)
      : "memory");
}
// the same with the (wave-uniform) destination handed over in a VECTOR register: under scalar-register pressure hipcc keeps such
// values in VGPRs and then fails to legalise an "s" asm operand (seen in duo_helper).  (s_nop first: the VALU instruction in front
// of the statement may have just written that register, and nothing pads the inside of an asm string.)
__device__ __forceinline__ void glds16_sc1_v(const void* gsrc, uint32_t lds_dst_uniform) {
  unsigned keep, dst;
  asm volatile(
      "s_nop 1\n\tv_readfirstlane_b32 %1, %3\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(dst)
      : "v"(gsrc), "v"(lds_dst_uniform)
      : "memory");
}
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ double ld8_sc1(const double* p) {  // compiler-tracked 8-byte load past the L1
  return __hip_atomic_load((const gdouble*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st16_sc1(double* p, v2d x) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");  // (data-register hazard)
}

// ---- dispatch ---------------------------------------------------------------------------------------------------
// Workers draw tasks from ONE list in order (atomicAdd on a head word) and wait for the flags of what they drew.  The
// list is a topological order (dag_build: the start order of a simulated launch), so whatever is at the head can run
// once the tasks before it have: no deadlock at any residency, no scheduler state.  The dispatchers tried before it
// (two lists, a claim window, dependency counters + ready queues with polling workers, a ticket FIFO, a side queue for
// urgent tasks) are recorded in tools/legacy_kernels/dag_dispatch_tickets.inc and DESIGN.md section 4.4.
constexpr uint32_t TASK_DONE = 0xfffffffeu, TASK_ERR = 0xfffffffdu;
// ctrl words: [0, 32) rare (arrival ticket, error, error info); the list head on a cache line of its own
constexpr int C_TICKET = 0, C_ERR = 2, C_ERRINFO = 3, C_HEAD = 32;
static_assert(DAG_CTRL_WORDS >= 64, "control block");

__device__ __forceinline__ uint32_t* dag_cnt(const DagArgs& a) { return a.ctrl + DAG_CTRL_WORDS; }  // [ntasks] start counts

// The workers' tile task (run_task) reads the launch arguments from the KERNARG segment with scalar loads.  (Rounds 3 / 4
// passed `const DagArgs&` to the noinline functions: the argument block then lived in scratch -- 42 scratch accesses per
// dispatched task, a second 120-byte copy per task of a batched launch, six flat loads at the head of every task.
// Measured in round 5, profiles/r05_dagk_ab.txt: N = 8192 update 7.42 -> 7.29 ms, a batched launch of fifteen 9.0 -> 8.2 ms,
// find_best_model_initialization(90) 62 -> 57 ms.)  run_chain keeps the reference: with the block in SGPRs the chain,
// which already fills 256 VGPRs, spills them.
static_assert(sizeof(DagArgs) == 120, "dag_args reads the block by offset");
// the launch arguments from the kernarg segment (explicit arguments start at offset 0): scalar loads only
__device__ __forceinline__ DagArgs dag_args(const void* kernarg) {
  const uint64_t p = (uint64_t)(uintptr_t)kernarg;
  const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(p >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)p);   // (the builtin returns a signed int)
  const __attribute__((address_space(4))) uint64_t* q = (const __attribute__((address_space(4))) uint64_t*)pu;
  DagArgs a;
  a.Ap = (double*)q[0];
  a.Lp = (double*)q[1];
  a.Wp = (double*)q[2];
  a.ld = (int64_t)q[3];
  a.NB = (int)(uint32_t)q[4];
  a.ntasks = (int)(uint32_t)(q[4] >> 32);
  a.tasks = (const DagTask*)q[5];
  a.chain_dep = (const uint32_t*)q[6];
  a.topo = (const uint32_t*)q[7];
  a.flags = (uint32_t*)q[8];
  a.ctrl = (uint32_t*)q[9];
  a.info = (int*)q[10];
  a.trace = (unsigned long long*)q[11];
  a.B = (int)(uint32_t)q[12];
  a.duo = (int)(uint32_t)(q[12] >> 32);
  a.mat_stride = (int64_t)q[13];
  a.flags_stride = (uint32_t)q[14];
  return a;
}

// run_chain works on a LOCAL copy of the argument block with every field made wave-uniform at entry.  (Through the
// reference the block lives in scratch and may alias every global store, so the compiler re-loaded fields inside the
// step loop -- 63 flat loads at loop depth 1, most of them dependent pairs on the chain's critical path, each behind an
// s_waitcnt vmcnt(0) lgkmcnt(0).  Measured in round 5, profiles/r05_dagk_ab.txt: N = 4096 update 1.88 -> 1.83 ms.)
#define TGP_DAG_CHAIN_INLINE __forceinline__
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint64_t u = (uint64_t)v;
  return (int64_t)(((uint64_t)uni((uint32_t)(u >> 32)) << 32) | uni((uint32_t)u));
}
__device__ __forceinline__ DagArgs dag_uniform_copy(const DagArgs& m) {
  DagArgs a;
  a.Ap = uniptr(m.Ap);
  a.Lp = uniptr(m.Lp);
  a.Wp = uniptr(m.Wp);
  a.ld = uni64(m.ld);
  a.NB = (int)uni((uint32_t)m.NB);
  a.ntasks = (int)uni((uint32_t)m.ntasks);
  a.tasks = uniptr(m.tasks);
  a.chain_dep = uniptr(m.chain_dep);
  a.topo = uniptr(m.topo);
  a.flags = uniptr(m.flags);
  a.ctrl = uniptr(m.ctrl);
  a.info = uniptr(m.info);
  a.trace = uniptr(m.trace);
  a.B = (int)uni((uint32_t)m.B);
  a.duo = (int)uni((uint32_t)m.duo);
  a.mat_stride = uni64(m.mat_stride);
  a.flags_stride = uni(m.flags_stride);
  return a;
}

// ---- generic tile task ----------------------------------------------------------------------------------------------
struct TaskU {  // a task descriptor with every field in scalar registers
  uint32_t a_off, b_off, c_off, o_off, nk, flags, a_mat, b_mat, c_mat, o_mat, set, sib;
};
// HALF (round 6, the split plan's T(i,i-2) and last burst of (i,i-1)): the task computes 64 of the tile's 128 rows -- rows 64 hi ..
// -- on 32 x 32 wave tiles (2 x 2 fragments instead of 4 x 2): half the A operand, half the MFMAs per wave, the same sum in the
// same order for every element.
template <bool HALF>
__device__ __attribute__((noinline)) void run_task(const void* kernarg, uint32_t mb, uint32_t idx) {
  const DagArgs a0 = dag_args(kernarg);
  const DagArgs a = a0.B > 1 ? dag_member(a0, (uint32_t)__builtin_amdgcn_readfirstlane(mb)) : a0;
  DAG_LDS_DECL;
  char* const lds = dag_lds;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;  // wave tile: rows 64 wr .. (HALF: 32 wr ..), columns 32 wc ..
  constexpr int RF = HALF ? 2 : 4;    // 16-row fragments per wave
  const DagTask* const tp = uniptr(a.tasks) + uni(idx);
  TaskU t;
  t.a_off = uni(tp->a_off); t.b_off = uni(tp->b_off); t.c_off = uni(tp->c_off); t.o_off = uni(tp->o_off);
  t.nk = uni(tp->nk); t.flags = uni(tp->flags);
  t.a_mat = uni(tp->a_mat); t.b_mat = uni(tp->b_mat); t.c_mat = uni(tp->c_mat); t.o_mat = uni(tp->o_mat);
  t.set = uni(tp->set);
  t.sib = uni(tp->dep3);
  double* const Ap = uniptr(a.Ap);
  double* const Lp = uniptr(a.Lp);
  double* const Wp = uniptr(a.Wp);
  auto mat = [&](uint32_t m) { return m == DAG_MAT_A ? Ap : (m == DAG_MAT_L ? Lp : Wp); };
  const int64_t ld = (int64_t)uni((uint32_t)a.ld);
  const bool nn = (t.flags & DAG_NN) != 0;
  const int64_t r0 = (HALF && (t.flags & DAG_HI)) ? (int64_t)(TILE / 2) * ld : 0;   // first row of the half, as an element offset
  const double* const Abase = mat(t.a_mat) + t.a_off + r0;
  const double* const Bbase = mat(t.b_mat) + t.b_off;
  const int64_t bstep = nn ? (int64_t)TILE * ld : TILE;  // from one k tile of B to the next
  const uint32_t lds0 = (uint32_t)(size_t)(lds_char*)lds;
  const int nchunks = (int)t.nk * (TILE / KC);

  // DMA sources of this lane: A group g = 4 w + u (rows 4 g + q), granule gp holds k = 2 ((gp - 4 q) & 15)
  const int q = lane >> 4, gp = lane & 15;
  const int64_t a_lane = (int64_t)q * ld + 2 * ((gp - 4 * q) & 15);
  const int64_t bn_lane = 2 * lane;  // NN: one k row per instruction, 64 lanes x 2 columns

  auto issue = [&](int c) {
    const int kt = c >> 2, kk = (c & 3) * KC;
    const uint32_t sa = lds0 + (uint32_t)((c & 1) * STAGE), sb = sa + STAGE_A;
    const double* const At = Abase + (int64_t)kt * TILE + kk;
#pragma unroll
    for (int u = 0; u < (HALF ? 2 : 4); ++u) {   // (HALF: 16 groups of four rows instead of 32)
      const int g = (HALF ? 2 : 4) * w + u;
      glds16_sc1(At + (int64_t)(4 * g) * ld + a_lane, sa + (uint32_t)(g * GROUP_B));
    }
    if (nn) {
      const double* const Bt = Bbase + (int64_t)kt * bstep + (int64_t)kk * ld;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = 4 * w + u;
        glds16_sc1(Bt + (int64_t)r * ld + bn_lane, sb + (uint32_t)(r * BN_ROW));
      }
    } else {
      const double* const Bt = Bbase + (int64_t)kt * TILE + kk;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int g = 4 * w + u;
        glds16_sc1(Bt + (int64_t)(4 * g) * ld + a_lane, sb + (uint32_t)(g * GROUP_B));
      }
    }
  };

  v4d acc[RF][2];
#pragma unroll
  for (int bi = 0; bi < RF; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = (v4d){0.0, 0.0, 0.0, 0.0};

  // operand read offsets of this lane inside a stage (row 16 b + lr, k = 4 k4 + lq)
  const int lane_rows = (lr >> 2) * GROUP_B + (lr & 3) * 256 + (lq & 1) * 8;
  const int c_lane = (lq >> 1) + 4 * (lr & 3);
  const int bn_off = lq * BN_ROW + lr * 8;

#ifndef TGP_DAG_DEFER
#define TGP_DAG_DEFER 1
#endif
  // Cross-barrier deferral (TGP_DAG_DEFER, round 6; the float64 sweep's form): the MFMAs of a chunk's LAST k step are issued
  // behind the next chunk's barrier -- their operands are in registers, the stage they came from may be overwritten -- so that the
  // matrix pipe has eight MFMAs per wave to run while the waves leave the barrier, request the next chunk and wait for the first
  // operand reads of this one.  Every accumulator still sees the same products in the same order: the same bits.
  double dav[RF], dbv[2];
#pragma unroll
  for (int bi = 0; bi < RF; ++bi) dav[bi] = 0.0;
  dbv[0] = dbv[1] = 0.0;
  auto deferred = [&]() {
#pragma unroll
    for (int bi = 0; bi < RF; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = mfma_f64(dav[bi], dbv[bj], acc[bi][bj]);
  };

  issue(0);
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    drain_vm();        // this wave's share of chunk c has landed ...
    __syncthreads();   // ... everyone's has, and everyone is done with the other stage (chunk c - 1)
    if (TGP_DAG_DEFER && c > 0) deferred();
    // Waves w and w + 4 share a SIMD: the first four request chunk c + 1 now, their partners after half of this chunk's
    // MFMAs -- eight LDS-DMA instructions cost a wave several hundred issue cycles, during which its partner feeds the
    // matrix pipe (TGP_DAG_STAGGER = 0: everybody up front, round 3)
    const bool issue_early = !TGP_DAG_STAGGER || w < 4;
    if (issue_early && c + 1 < nchunks) issue(c + 1);
    const char* const sa = lds + (c & 1) * STAGE;
    const char* const sb = sa + STAGE_A;
    // operands of k step k4 + 1 are fetched BEFORE the MFMAs of step k4 are issued (the scheduling barriers keep hipcc
    // from sinking the reads back below them): with two waves per SIMD an exposed LDS round trip per k step costs ~10 %
    double av[2][RF], bv[2][2];
    auto fetch = [&](int k4, double (&fa)[RF], double (&fb)[2]) {
      const int rot = ((2 * k4 + c_lane) & 15) << 4;
#pragma unroll
      for (int bi = 0; bi < RF; ++bi) fa[bi] = *(const double*)(sa + (4 * RF * wr + 4 * bi) * GROUP_B + lane_rows + rot);
      if (nn) {
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) fb[bj] = *(const double*)(sb + 4 * k4 * BN_ROW + bn_off + (32 * wc + 16 * bj) * 8);
      } else {
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) fb[bj] = *(const double*)(sb + (8 * wc + 4 * bj) * GROUP_B + lane_rows + rot);
      }
    };
    fetch(0, av[0], bv[0]);
    if (TGP_DAG_SETPRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int k4 = 0; k4 < KC / 4; ++k4) {
      if (k4 + 1 < KC / 4) fetch(k4 + 1, av[(k4 + 1) & 1], bv[(k4 + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (TGP_DAG_DEFER && k4 == KC / 4 - 1) {
#pragma unroll
        for (int bi = 0; bi < RF; ++bi) dav[bi] = av[k4 & 1][bi];
        dbv[0] = bv[k4 & 1][0];
        dbv[1] = bv[k4 & 1][1];
      } else {
#pragma unroll
        for (int bi = 0; bi < RF; ++bi)
#pragma unroll
          for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = mfma_f64(av[k4 & 1][bi], bv[k4 & 1][bj], acc[bi][bj]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k4 == TGP_DAG_LATE_K4 && !issue_early && c + 1 < nchunks) issue(c + 1);
    }
    if (TGP_DAG_SETPRIO) __builtin_amdgcn_s_setprio(0);
  }
  if (TGP_DAG_DEFER) deferred();
  __syncthreads();  // the stages are dead: the tile goes through LDS once, so that global traffic is 16 B per lane
  double* const T = (double*)lds;
#pragma unroll
  for (int bi = 0; bi < RF; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        T[(16 * RF * wr + 16 * bi + lq + 4 * r) * OUT_LD + 32 * wc + 16 * bj + lr] = acc[bi][bj][r];
  __syncthreads();
  {
    const double alpha = (t.flags & DAG_NEG) ? -1.0 : 1.0;
    const bool beta = (t.flags & DAG_BETA) != 0;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(mat(t.c_mat), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(mat(t.o_mat), 0, 0x7fffffff, 0x00020000);
    constexpr int NIT = (HALF ? TILE / 2 : TILE) * TILE / 1024;  // 16 (HALF: 8) passes of 512 threads x 2 doubles
    // (measured in round 6 and rejected: requesting the C tile two chunks before the end of the k loop instead of here -- 48
    // callee-saved registers spilled at the task's entry and every task 1 - 1.5 us SLOWER, N = 8192 7.3 -> 7.65 ms: the epilogue
    // is bound by its write-through stores, not by this round trip)
    v2d cin[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 512 * it, row = e >> 6, c2 = (e & 63) * 2;
      cin[it] = (v2d){0.0, 0.0};
      if (beta) {
        const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rc, (int)(((int64_t)t.c_off + r0 + (int64_t)row * ld + c2) * 8), 0, 16);
        cin[it] = __builtin_bit_cast(v2d, raw);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 512 * it, row = e >> 6, c2 = (e & 63) * 2;
      const v2d v = *(const v2d*)(T + row * OUT_LD + c2);
      v2d o;
      o.x = fma(alpha, v.x, cin[it].x);
      o.y = fma(alpha, v.y, cin[it].y);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), ro,
                                             (int)(((int64_t)t.o_off + r0 + (int64_t)row * ld + c2) * 8), 0, 16);
    }
  }
  drain_vm();  // every storing wave drains its write-through stores, THEN the barrier, THEN one lane publishes
  __syncthreads();
  if (tid == 0) {
    // DAG_SIB: this flag also vouches for the lower half, a task earlier in the list (running, or done long ago)
    if (HALF && (t.flags & DAG_SIB)) wait_flag_at(uniptr(a.flags), uniptr(a.ctrl), t.sib);
    st_flag(uniptr(a.flags) + t.set, 1u);
  }
}

// ---- the chain workgroup ------------------------------------------------------------------------------------------
__device__ TGP_DAG_CHAIN_INLINE bool chain_wait(const DagArgs& a, uint32_t id, volatile uint32_t* ctl) {
  if (threadIdx.x == 0) ctl[1] = wait_flag(a, id) ? 1u : 0u;
  __syncthreads();
  const bool ok = ctl[1] != 0;
  __syncthreads();
  return ok;
}

// pass `it` (0..15) of the copy of Lsub = L(j,j-1) from LDS to global memory: 512 threads x 16 B, write-through
__device__ __forceinline__ v2d push_lsub_read(const double* S, int it) {
  const int e = threadIdx.x + 512 * it, i = e >> 6, c2 = (e & 63) * 2;
  return *(const v2d*)(S + i * QS + c2);
}
__device__ __forceinline__ void push_lsub_store(double* Ls, int64_t ld, v2d x, int it) {
  const int e = threadIdx.x + 512 * it, i = e >> 6, c2 = (e & 63) * 2;
  st16_sc1(Ls + (int64_t)i * ld + c2, x);
}
__device__ __forceinline__ void push_lsub_pass(double* Ls, int64_t ld, const double* S, int it) {
  push_lsub_store(Ls, ld, push_lsub_read(S, it), it);
}

// Step j, part 1: S (LDS) = lower triangle of the diagonal tile with every earlier column subtracted.
//   j == 0: the tile as the assembly kernel left it (nothing can be pending);
//   j  > 0: P(j,j) - Lsub Lsub^T with Lsub = L(j,j-1) row-major in LDS (left there by part 3 of step j - 1).
// `pending`: the flag of L(j,j-1), which so far exists in LDS only: its 16 store passes are interleaved with the first 16
// k steps of this product and the flag goes out at the product's barrier (NONE: already stored and published).
__device__ __forceinline__ void chain_diag(const DagArgs& a, int j, uint32_t pending) {
  DAG_LDS_DECL;
  double* const S = (double*)dag_lds;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const S8 = (const char*)S;
  const int lane_a = (lr * QS + lq) * 8, lane_o = (lq * QS + lr) * 8;
  j = __builtin_amdgcn_readfirstlane(j);
  const int64_t ld = (int64_t)uni((uint32_t)a.ld), off = (int64_t)j * TILE;
  const double* const Pd = uniptr(a.Ap) + off * ld + off;  // tile (j, j)
  const int ld32 = (int)ld;  // (offsets inside a tile fit 32 bits: one address register per load instead of two)
  if (j == 0) {
    constexpr int NIT = QN * QN / 1024;
    v2d x[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 512 * it, i = e >> 6, c2 = (e & 63) * 2;
      x[it] = (v2d){0.0, 0.0};
      if (c2 <= i) x[it] = *(const v2d*)(Pd + (int64_t)i * ld + c2);  // written by the launch before this one
      if (c2 + 1 > i) x[it].y = 0.0;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 512 * it, i = e >> 6, c2 = (e & 63) * 2;
      *(v2d*)(S + i * QS + c2) = x[it];
    }
    __syncthreads();
    return;
  }
  // The 36 lower fragments: block rows p and 7 - p make 9 fragments, n <= p: (p, n), n > p: (7 - p, n - p - 1); the
  // pair of waves 2 p, 2 p + 1 splits them by parity (5 + 4).  Everything about a fragment is wave-uniform.
  constexpr int NF = 5;
  // (waves w and w + 4 share a SIMD: one of them takes the five-fragment half of its pair, the other the four-fragment
  // half -- nine fragments per SIMD instead of ten and eight)
  const int p = w >> 1, h = (w ^ (w >> 2)) & 1;
  v4d acc[NF];
  double pin[NF][4];
  int fbi[NF], fbj[NF];
  bool live[NF];
#pragma unroll
  for (int m = 0; m < NF; ++m) {
    const int n = 2 * m + h;
    live[m] = n < 9;
    fbi[m] = n <= p ? p : 7 - p;
    fbj[m] = n <= p ? n : (live[m] ? n - p - 1 : 0);
  }
#pragma unroll
  for (int m = 0; m < NF; ++m) {
    acc[m] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) pin[m][r] = ld8_sc1(Pd + ((16 * fbi[m] + lq + 4 * r) * ld32 + 16 * fbj[m] + lr));
  }
  double* const Lprev = uniptr(a.Lp) + off * ld + (off - TILE);  // L(j, j-1)
  const bool push = pending != NONE;
  // 32 k steps; the operands of step t + 1 are fetched BEFORE the MFMAs of step t are issued (by hand: the asm stores
  // of the interleaved copy are compiler barriers, hipcc would not hoist the reads across them)
  double a_lo, a_hi, bv[NF];
  v2d pv = {0.0, 0.0};
  auto fetch = [&](int t, double& lo, double& hi, double (&b)[NF]) {
    const int kb = t >> 2, k4 = t & 3;
    lo = ldsd(S8 + lane_a + blk(p, kb) + 32 * k4);        // the wave's two block rows
    hi = ldsd(S8 + lane_a + blk(7 - p, kb) + 32 * k4);
#pragma unroll
    for (int m = 0; m < NF; ++m) b[m] = ldsd(S8 + lane_a + blk(fbj[m], kb) + 32 * k4);
  };
  fetch(0, a_lo, a_hi, bv);
#pragma unroll
  for (int t = 0; t < 4 * QB; ++t) {
    double n_lo = 0.0, n_hi = 0.0, nb[NF];
    if (t + 1 < 4 * QB) fetch(t + 1, n_lo, n_hi, nb);
    // 16 store passes over 32 steps (under the ~20 GB/s a CU's write-through stores drain at); a pass reads its 16 bytes
    // from LDS in an even step and stores them in the next one: the store never waits for an LDS round trip
    if (push && (t & 1) == 0) pv = push_lsub_read(S, t >> 1);
    if (push && (t & 1) == 1) push_lsub_store(Lprev, ld, pv, t >> 1);
#pragma unroll
    for (int m = 0; m < NF; ++m) acc[m] = mfma_f64((2 * m + h) <= p ? a_lo : a_hi, bv[m], acc[m]);
    if (t + 1 < 4 * QB) {
      a_lo = n_lo;
      a_hi = n_hi;
#pragma unroll
      for (int m = 0; m < NF; ++m) bv[m] = nb[m];
    }
  }
  stamp((a.trace && tid == 0) ? a.trace + CT * j + 6 : nullptr);
  stamp((a.trace && lane == 0) ? a.trace + CT * j + 24 + w : nullptr);
  drain_vm();       // (the write-through stores of L(j,j-1) issued before this product: long landed)
  __syncthreads();  // everybody is done reading Lsub: S takes its place
  if (tid == 0 && pending != NONE) {
    st_flag(uniptr(a.flags) + pending, 1u);
  }
#pragma unroll
  for (int m = 0; m < NF; ++m) {
    if (!live[m]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * fbi[m] + lq + 4 * r, col = 16 * fbj[m] + lr;
      const double v = pin[m][r] - acc[m][r];
      *(double*)((char*)S + lane_o + blk(fbi[m], fbj[m]) + 4 * RB * r) = (col <= row) ? v : 0.0;
    }
  }
  drain_vm();  // (lane 0's flag store and event count: no wave enters the leaf with a memory wait still ahead of it)
  __syncthreads();
}

// Step j, part 2: the 128-leaf on S (L_jj goes to global memory panel by panel, write-through), then W_jj = T.
// `peek`: the flag the chain waits for next -- lane 0 looks at it before this part's closing barrier, which then also
// hands the answer round (returned: the flag was up, the usual case; saves the two barriers of a wait of its own).
__device__ __forceinline__ bool chain_leaf(const DagArgs& a, int j, uint32_t peek, uint32_t peek2 = NONE) {
  DAG_LDS_DECL;
  double* const S = (double*)dag_lds;
  const WorkItem* const items = (const WorkItem*)(S + QN * QS);
  const int tid = threadIdx.x;
  j = __builtin_amdgcn_readfirstlane(j);
  const int64_t ld = (int64_t)uni((uint32_t)a.ld), off = (int64_t)j * TILE;
  double* const Wp = uniptr(a.Wp);
  // (measured and rejected: a worker wave pulling P(j+1,j) into the L2 during the leaf with dummy LDS-DMA loads -- the
  // leaf slows down by 2.5 us and part 3's operand loads do not get faster: they are address-rate bound, not misses)
  leaf_core<true>(S, items, uniptr(a.Lp), ld, off, uniptr(a.info), (lds_sync_t*)(lds_char*)(dag_lds + CTL_OFF + 32),
                  Wp + off * ld + off);  // (stores W_jj as it goes)
  drain_vm();
  volatile uint32_t* const ctl = (volatile uint32_t*)(dag_lds + CTL_OFF);
  if (tid == 0)
    ctl[1] = ((peek == NONE || ld_flag(uniptr(a.flags) + peek) != 0) && (peek2 == NONE || ld_flag(uniptr(a.flags) + peek2) != 0)) ? 1u : 0u;
  __syncthreads();
  return ctl[1] != 0;
}

// Step j, part 3: L(j+1,j) = P(j+1,j) W_jj^T.  Wave w owns rows 16 w .. of the tile, held transposed as natural B
// operands; W_jj is read from S; the result replaces it there (row-major: the operand of the next step's part 1) and
// goes to global memory write-through.
__device__ __forceinline__ bool chain_sub(const DagArgs& a, int j, uint32_t peek) {
  DAG_LDS_DECL;
  double* const S = (double*)dag_lds;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const S8 = (const char*)S;
  const int lane_a = (lr * QS + lq) * 8;
  j = __builtin_amdgcn_readfirstlane(j);
  const int64_t ld = (int64_t)uni((uint32_t)a.ld), off = (int64_t)j * TILE;
  const double* const Ps = uniptr(a.Ap) + (off + TILE) * ld + off;
  // The MFMA sums over k whatever order the two operands agree on: inside a 16-block lane (lq, lr) takes
  // k = 4 lq + k4 (not 4 k4 + lq), so that its four B values P[16 w + lr][16 kc + 4 lq .. + 3] are 32 contiguous bytes
  // -- two 16-byte loads per block instead of four scattered 8-byte ones (the operand loads of this product are
  // address-rate bound and sit ON the chain).  The A operand W_jj(kb, kc) comes from LDS with the same k.
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((double*)Ps, 0, 0x7fffffff, 0x00020000);
  double pb[QB][4];
#pragma unroll
  for (int kc = 0; kc < QB; ++kc) {
    const int voff = ((16 * w + lr) * (int)ld + 16 * kc + 4 * lq) * 8;
    const v4u lo = __builtin_amdgcn_raw_buffer_load_b128(rp, voff, 0, 16);
    const v4u hi = __builtin_amdgcn_raw_buffer_load_b128(rp, voff + 16, 0, 16);
    const v2d l2 = __builtin_bit_cast(v2d, lo), h2 = __builtin_bit_cast(v2d, hi);
    pb[kc][0] = l2.x; pb[kc][1] = l2.y; pb[kc][2] = h2.x; pb[kc][3] = h2.y;
  }
  if (a.trace) {  // development aid: when did this wave's operand loads land?
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(lane == 0 ? a.trace + CT * j + 16 + w : nullptr);
  }
  const int lane_ap = (lr * QS + 4 * lq) * 8;  // A operand with k = 4 lq + k4: + 8 k4
  v4d o[QB];
#pragma unroll
  for (int kb = 0; kb < QB; ++kb) {
    o[kb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc <= kb; ++kc)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) o[kb] = mfma_f64(ldsd(S8 + lane_ap + blk(kb, kc) + 8 * k4), pb[kc][k4], o[kb]);
  }
  stamp((a.trace && tid == 0) ? a.trace + CT * j + 7 : nullptr);
  stamp((a.trace && lane == 0) ? a.trace + CT * j + 8 + w : nullptr);
  __syncthreads();  // W_jj has been read (and stored): S becomes Lsub, row-major
#pragma unroll
  for (int kb = 0; kb < QB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(16 * w + lr) * QS + 16 * kb + lq + 4 * r] = o[kb][r];
  volatile uint32_t* const ctl = (volatile uint32_t*)(dag_lds + CTL_OFF);
  if (tid == 0) ctl[1] = (peek == NONE || ld_flag(uniptr(a.flags) + peek) != 0) ? 1u : 0u;
  __syncthreads();
  // L(j+1,j) goes to global memory from here (LDS) UNDERNEATH the next step's diagonal product (chain_diag), which
  // also publishes its flag -- pushing 128 KB of write-through stores costs this CU ~6 us when nothing hides it
  return ctl[1] != 0;
}

__device__ __attribute__((noinline)) void run_chain(const DagArgs& a_mem) {
  const DagArgs a = dag_uniform_copy(a_mem);
  DAG_LDS_DECL;
  char* const lds = dag_lds;
  double* const S = (double*)lds;
  WorkItem* const items = (WorkItem*)(S + QN * QS);
  volatile uint32_t* const ctl = (volatile uint32_t*)(lds + CTL_OFF);
  const int tid = threadIdx.x;
  leaf_build_items(items, tid);
  if (tid == 0) *(lds_sync_t*)(lds_char*)(dag_lds + CTL_OFF + 32) = 0;  // the panel waves' hand-shake word (tgp_leaf_dev.inc)
  const uint32_t WD = (uint32_t)a.ntasks, LSUB = (uint32_t)(a.ntasks + a.NB);
  uint32_t pending = NONE;  // flag of L(j,j-1): stores issued, not yet drained / published
  // Before BLOCKING on a flag everything this workgroup owes must be out (whoever sets that flag may be waiting for
  // it); when the flag is already up -- the usual case -- the pending publication stays deferred.
  auto wait_for = [&](uint32_t id) -> bool {
    if (id == NONE) return true;
    if (tid == 0) ctl[1] = ld_flag(a.flags + id) != 0 ? 1u : 0u;
    __syncthreads();
    const bool up = ctl[1] != 0;
    __syncthreads();
    if (up) return true;
    if (pending != NONE) {  // L(j,j-1) is still in LDS only: store it now
      const int jj = (int)(pending - LSUB) + 1;
      double* const Lprev = a.Lp + (int64_t)jj * TILE * a.ld + (int64_t)(jj - 1) * TILE;
      for (int it = 0; it < QN * QN / 1024; ++it) push_lsub_pass(Lprev, a.ld, S, it);
      drain_vm();
      __syncthreads();
      if (tid == 0) st_flag(a.flags + pending, 1u);
      pending = NONE;
    }
    return chain_wait(a, id, ctl);
  };
  bool diag_up = false;  // the flag step j's diagonal product needs was seen up at the end of step j - 1
#pragma unroll 1
  for (int j = 0; j < a.NB; ++j) {
    unsigned long long* const tr = (a.trace && tid == 0) ? a.trace + CT * j : nullptr;
    stamp(tr);
    if (!diag_up && !wait_for(a.chain_dep[2 * j])) return;
    stamp(tr ? tr + 1 : nullptr);
    chain_diag(a, j, pending);
    pending = NONE;
    stamp(tr ? tr + 2 : nullptr);
    const bool last = j + 1 == a.NB;
    // (the split plan finishes P(j+1,j) in two half-tile tasks: a second flag, chain_dep[2 NB + j])
    const uint32_t sub_dep = last ? NONE : a.chain_dep[2 * j + 1], sub_dep2 = last ? NONE : a.chain_dep[2 * a.NB + j];
    const bool sub_up = chain_leaf(a, j, sub_dep, sub_dep2);
    if (tid == 0) st_flag(a.flags + WD + j, 1u);
    stamp(tr ? tr + 3 : nullptr);
    if (last) break;
    if (!sub_up && !(wait_for(sub_dep) && wait_for(sub_dep2))) return;
    stamp(tr ? tr + 4 : nullptr);
    diag_up = chain_sub(a, j, a.chain_dep[2 * j + 2]);
    pending = LSUB + (uint32_t)j;
    stamp(tr ? tr + 5 : nullptr);
  }
}

// ---- round 6: the chain as TWO workgroups ("duo") --------------------------------------------------------------------
// run_chain's step is diag 13.4 + leaf 26.1 + sub 14.2 us on ONE compute unit, and during the leaf that unit's matrix pipes are
// nearly idle.  Here the two products move to a SECOND workgroup that runs them UNDERNEATH the other one's leaf, panel by panel,
// and the two workgroups swap roles every step, so that the diagonal block never crosses compute units:
//   workgroup c = j & 1 runs leaf(j); meanwhile workgroup 1 - c is the HELPER of step j + 1: it holds P(j+1,j) in registers
//   (wave w: rows 16 w .., as eight transposed 16 x 16 accumulator blocks) and the 36 lower fragments of L(j+1,j) L(j+1,j)^T in
//   accumulators, and for every panel kc of leaf(j) -- published through three flag words: by the leaf's two panel waves once their
//   write-through stores of the L panel have drained (a panel late, where the wait is free: tgp_leaf_dev.inc panel_factor), and by
//   the wave that idles in [C kb] for W_d(kc) = L_d(kc)^-1, which it pushes --
//       L(j+1,j)[:, kc]   = X[:, kc] W_d(kc)^T                        (4 MFMAs per wave)
//       X[:, kb]         -= L(j+1,j)[:, kc] L_jj(kb, kc)^T,  kb > kc   (right-looking: 4 (7 - kc) MFMAs)
//       fragments        += L(j+1,j)[:, kc] L(j+1,j)[:, kc]^T          (the 36 lower fragments: 16 - 20 MFMAs)
//   and at the end S(j+1,j+1) = P(j+1,j+1) - fragments (the diagonal tile is asked for two blocks before the end: its last product
//   ends later than P(j+1,j)); it then runs leaf(j+1) itself on that S -- while the other workgroup becomes the helper of step j + 2.
// What crosses compute units per step is the stream of panels (<= 16 KiB each, through the L2 like every other tile, LDS-DMA'd as
// soon as they are out -- all of them up front when the helper starts behind the leaf, which is the usual case).  Measured
// (profiles/r06_update_breakdown.txt): the helper has P(j+1,j) 16.7 us into the other workgroup's leaf, needs 6.8 us of loads and 25 us
// for its eight column blocks (15.4 us of MFMA time), the step is 49.5 us against run_chain's 54 -- and tools/dag_exec_sim.py shows
// why not less: below ~49 us the first half of the launch is bound by tile-task throughput.
// L(j+1,j) is formed as a blocked triangular solve against L_jj (inverted 16-blocks) rather than as the product with W_jj: the
// same matrix to rounding, NOT the same bits as run_chain's (tests/test_gpu_dag.py compares the two to a tolerance); the
// arithmetic is still fixed by the plan alone: bit-identical run to run, handle to handle, on any share of the GPU.
// Inside a 16-block the contraction index and the accumulator row are permuted by sigma(i) = 4 (i & 3) + (i >> 2) so that
// every lane's four operand values are 32 contiguous bytes (in global memory and in LDS alike).
constexpr int DUO_PANEL_B = QN * 16 * 8;  // 16 KiB: a staged panel [W_d: 16 rows][L_jj rows below] or a column block of L(j+1,j): [128][16]
constexpr int DUO_PF = DAG_DUO_PF;        // panel flag words per step: [panel][panel wave 0, panel wave 1, W_d's wave, -]
// LDS of the helper (all inside what becomes the leaf's S): the eight staged panels back to back (panel kc: 128 - 16 kc rows of 128
// bytes), then two column-block buffers
__host__ __device__ constexpr int duo_poff(int kc) { return 128 * (128 * kc - 8 * kc * (kc - 1)); }
constexpr int DUO_LP_OFF = duo_poff(QB);  // 73 728
static_assert(DUO_LP_OFF + 2 * DUO_PANEL_B <= QN * QS * 8, "the helper's buffers live where the leaf's S will be");

__device__ __forceinline__ uint32_t* duo_panel_flags(const DagArgs& a) { return a.ctrl + DAG_CTRL_WORDS + a.ntasks; }

// The helper of step j >= 1: L(j,j-1) -> global memory (flag LSUB + j - 1), S(j,j) -> LDS, ready for leaf(j).  False: a wait
// failed (every thread returns the same).
__device__ __forceinline__ bool duo_helper(const DagArgs& a, int j) {
  DAG_LDS_DECL;
  char* const lds = dag_lds;
  double* const S = (double*)lds;
  volatile uint32_t* const ctl = (volatile uint32_t*)(lds + CTL_OFF);
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  j = __builtin_amdgcn_readfirstlane(j);
  const int64_t ld = a.ld, off = (int64_t)j * TILE, offp = off - TILE;
  const int ld32 = (int)ld;
  unsigned long long* const tr = (a.trace && tid == 0) ? a.trace + CT * j : nullptr;
  stamp(tr);
  const uint32_t* const pf = duo_panel_flags(a) + DUO_PF * (j - 1);
  // thread 0: how many panels of leaf(j-1) are out (all three words up), counted from the first
  auto panels_out = [&]() -> uint32_t {
    uint32_t f[DUO_PF], n = 0;
#pragma unroll
    for (int i = 0; i < DUO_PF; ++i) f[i] = (i & 3) == 3 ? 1u : ld_flag(pf + i);
    bool run = true;
#pragma unroll
    for (int k = 0; k < QB; ++k) {
      run = run && f[4 * k] != 0 && f[4 * k + 1] != 0 && f[4 * k + 2] != 0;
      n += run ? 1u : 0u;
    }
    return n;
  };
  // thread 0, blocking: the bulk's three flags (first call) and panel kc -- everything polled TOGETHER, one round trip per look
  // (one after the other these were seven dependent round trips of ~0.7 us at the head of every helper); -> the number of
  // panels that are out, 0 on a timeout or when another workgroup has raised the error word
  auto wait_inputs = [&](int kc, bool bulk) -> uint32_t {
    // (the diagonal tile's own flag, chain_dep[2 j], is NOT waited for here: its last product -- a whole tile task that starts when
    // L(j,j-2) is out -- ends ~18 us into the other workgroup's leaf, later than P(j,j-1); the tile is only needed at the very end)
    const uint32_t d0 = bulk ? a.chain_dep[2 * j - 1] : NONE, d1 = bulk ? a.chain_dep[2 * a.NB + j - 1] : NONE;
    unsigned spins = 0;
    for (;;) {
      const uint32_t f0 = d0 == NONE ? 1u : ld_flag(a.flags + d0), f1 = d1 == NONE ? 1u : ld_flag(a.flags + d1);
      const uint32_t n = panels_out();
      if (f0 != 0 && f1 != 0 && n > (uint32_t)kc) return n;
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 255u) == 0) {
        if (ld_flag(a.ctrl + 2) != 0) return 0u;
        if (spins > SPIN_LIMIT) {
          st_flag(a.ctrl + 2, DAG_ERR_TIMEOUT);
          st_flag(a.ctrl + 3, 0x40000000u | ((uint32_t)j << 8) | (uint32_t)kc);
          return 0u;
        }
      }
    }
  };
  if (tid == 0) ctl[4] = wait_inputs(0, true);
  __syncthreads();
  if (ctl[4] == 0) return false;
  uint32_t avail = ctl[4];  // panels 0 .. avail - 1 are out
  stamp(tr ? tr + 1 : nullptr);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((double*)a.Ap, 0, 0x7fffffff, 0x00020000);
  const int tile_sub = (int)((off * ld + offp) * 8);  // byte offset of tile (j,j-1)
  const int tile_dd = (int)((off * ld + off) * 8);    // ... of tile (j,j)
  // Staging of the panels [k0, k1) of leaf(j-1) by LDS-DMA (no registers; the issuing wave drains before the barrier that hands
  // them round): panel kc = [128 - 16 kc rows][16 doubles], rows 0 .. 15 W_d(kc) from W's diagonal tile, then L_jj's rows below;
  // a wave's 64 lanes x 16 bytes are eight rows
  const uint32_t lds0 = (uint32_t)(size_t)(lds_char*)lds;
  const double* const Wt = a.Wp + offp * ld + offp;
  const double* const Lt = a.Lp + offp * ld + offp;
  auto stage_dma = [&](int k0, int k1) {
#pragma unroll
    for (int kc = 0; kc < QB; ++kc) {
      if (kc < k0 || kc >= k1) continue;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row0 = 64 * it + 8 * w;  // (wave-uniform: inside the panel or not)
        if (row0 < QN - 16 * kc) {
          // (rows of 128 bytes one behind the other put every 16-lane group of a 16-byte operand read on TWO bank groups -- 8-way
          // conflicts, measured as 25 us of column blocks against 15 of MFMA time; so LDS granule g of row R holds the row's
          // granule g ^ ((R >> 1) & 7): the lanes lr = 0 .. 15 of such a read then cover all 64 banks once)
          const int row = row0 + (lane >> 3), c2 = (((lane & 7) ^ (row >> 1)) & 7) * 2;
          const double* const src = ((it == 0 && w < 2) ? Wt : Lt) + (int64_t)(16 * kc + row) * ld + 16 * kc + c2;
          glds16_sc1_v(src, lds0 + (uint32_t)(duo_poff(kc) + row0 * 128));
        }
      }
    }
  };
  // the panels that are out first (untracked asm loads: behind hipcc's own they would make every counted wait of its an over-wait),
  // then X block by block -- block 0's products start while the rest of X streams in -- then the diagonal tile's fragments
  stage_dma(0, (int)avail);
  // X = P(j,j-1), rows 16 w ..: xt[kb][r] = X[16 w + lr][16 kb + 4 lq + r]
  v4d xt[QB];
#pragma unroll
  for (int kb = 0; kb < QB; ++kb) {
    const int voff = ((16 * w + lr) * ld32 + 16 * kb + 4 * lq) * 8;
    const v2d l2 = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, tile_sub, 16));
    const v2d h2 = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rA, voff + 16, tile_sub, 16));
    xt[kb] = (v4d){l2.x, l2.y, h2.x, h2.y};
  }
  // the 36 lower fragments of the diagonal tile, handed out as in chain_diag; sacc = the sum of the column blocks' squares, and
  // pin = P(j,j), requested two blocks before the end (S = pin - sacc)
  constexpr int NF = 5;
  const int p = w >> 1, h = (w ^ (w >> 2)) & 1;
  v4d sacc[NF];
  double pin[NF][4];
  int fbi[NF], fbj[NF];
  bool live[NF];
  const double* const Pd = a.Ap + off * ld + off;
#pragma unroll
  for (int m = 0; m < NF; ++m) {
    const int n = 2 * m + h;
    live[m] = n < 9;
    fbi[m] = n <= p ? p : 7 - p;
    fbj[m] = n <= p ? n : (live[m] ? n - p - 1 : 0);
    sacc[m] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) pin[m][r] = 0.0;
  }
  // (everything older than X's first block -- the panels' DMA -- has landed once that block is in registers: loads return in order)
  asm volatile("" : "+v"(xt[0]));
  uint32_t staged = avail;  // panels 0 .. staged - 1 are (being) staged; a wave's own DMA is drained before the next barrier it matters for
  __syncthreads();
  double* const Lg = a.Lp + off * ld + offp + (int64_t)(16 * w + lr) * ld + 4 * lq;  // this lane's 32 bytes of a column block
  const int sg = 4 * (lr & 3) + (lr >> 2);                                            // sigma(lr)
  // byte offsets of this lane's two 16-byte granules (columns 4 lq .. + 1, + 2 .. + 3) in a staged panel's row sigma(lr) and in a
  // column-block buffer's row lr (the swizzle of stage_dma)
  const int arow = sg * 128 + (((2 * lq) ^ (sg >> 1)) & 7) * 16, arow1 = sg * 128 + (((2 * lq + 1) ^ (sg >> 1)) & 7) * 16;
  const int lrow = lr * 128 + (((2 * lq) ^ (lr >> 1)) & 7) * 16, lrow1 = lr * 128 + (((2 * lq + 1) ^ (lr >> 1)) & 7) * 16;
  // column block kc of L(j,j-1): D[i][c] = L[16 w + c][16 kc + sigma(i)] = sum_k W_d[sigma(i)][sigma(k)] X[16 w + c][16 kc + sigma(k)]
  // (two chains of two: a dependent float64 MFMA issues only every ~128 cycles) -> LDS (for the fragment products) and global memory
  auto col_block = [&](int kc) -> v4d {
    const char* const st = lds + duo_poff(kc);
    char* const lp = lds + DUO_LP_OFF + (kc & 1) * DUO_PANEL_B;
    const v2d a01 = *(const v2d*)(st + arow), a23 = *(const v2d*)(st + arow1);
    v4d l0 = mfma_f64(a01.x, xt[kc][0], (v4d){0.0, 0.0, 0.0, 0.0});
    v4d l1 = mfma_f64(a01.y, xt[kc][1], (v4d){0.0, 0.0, 0.0, 0.0});
    l0 = mfma_f64(a23.x, xt[kc][2], l0);
    l1 = mfma_f64(a23.y, xt[kc][3], l1);
    const v4d lt = l0 + l1;
    const v2d o01 = (v2d){lt[0], lt[1]}, o23 = (v2d){lt[2], lt[3]};
    *(v2d*)(lp + w * 2048 + lrow) = o01;
    *(v2d*)(lp + w * 2048 + lrow1) = o23;
    st16_sc1(Lg + 16 * kc, o01);
    st16_sc1(Lg + 16 * kc + 2, o23);
    return lt;
  };
  // X[:, kb] -= L[:, kc] L_jj(kb, kc)^T, kb > kc: the k step outside, the independent blocks inside
  auto trailing = [&](int kc, v4d lt) {
    const char* const st = lds + duo_poff(kc);
    const v4d nlt = -lt;
    v2d b01[QB], b23[QB];
#pragma unroll
    for (int kb = 0; kb < QB; ++kb) {
      if (kb <= kc) continue;
      b01[kb] = *(const v2d*)(st + arow + (kb - kc) * 2048);
      b23[kb] = *(const v2d*)(st + arow1 + (kb - kc) * 2048);
    }
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
      if (kb > kc) xt[kb] = mfma_f64(b01[kb].x, nlt[0], xt[kb]);
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
      if (kb > kc) xt[kb] = mfma_f64(b01[kb].y, nlt[1], xt[kb]);
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
      if (kb > kc) xt[kb] = mfma_f64(b23[kb].x, nlt[2], xt[kb]);
#pragma unroll
    for (int kb = 0; kb < QB; ++kb)
      if (kb > kc) xt[kb] = mfma_f64(b23[kb].y, nlt[3], xt[kb]);
  };
  // -S(j,j) += L[:, kc] L[:, kc]^T on this wave's fragments
  auto fragments = [&](int kc) {
    const char* const lp = lds + DUO_LP_OFF + (kc & 1) * DUO_PANEL_B;
    const v2d lo01 = *(const v2d*)(lp + lrow + p * 2048), lo23 = *(const v2d*)(lp + lrow1 + p * 2048);
    const v2d hi01 = *(const v2d*)(lp + lrow + (7 - p) * 2048), hi23 = *(const v2d*)(lp + lrow1 + (7 - p) * 2048);
    v2d b01[NF], b23[NF];
#pragma unroll
    for (int m = 0; m < NF; ++m) {
      b01[m] = *(const v2d*)(lp + lrow + fbj[m] * 2048);
      b23[m] = *(const v2d*)(lp + lrow1 + fbj[m] * 2048);
    }
#pragma unroll
    for (int m = 0; m < NF; ++m)
      if (live[m]) sacc[m] = mfma_f64((2 * m + h) <= p ? lo01.x : hi01.x, b01[m].x, sacc[m]);
#pragma unroll
    for (int m = 0; m < NF; ++m)
      if (live[m]) sacc[m] = mfma_f64((2 * m + h) <= p ? lo01.y : hi01.y, b01[m].y, sacc[m]);
#pragma unroll
    for (int m = 0; m < NF; ++m)
      if (live[m]) sacc[m] = mfma_f64((2 * m + h) <= p ? lo23.x : hi23.x, b23[m].x, sacc[m]);
#pragma unroll
    for (int m = 0; m < NF; ++m)
      if (live[m]) sacc[m] = mfma_f64((2 * m + h) <= p ? lo23.y : hi23.y, b23[m].y, sacc[m]);
  };
  // Software pipeline: column block kc + 1 (a dependent chain: operand read, two MFMA levels, an add, the LDS write) is formed
  // BEFORE the fragment products of block kc, which fill the matrix pipe meanwhile.  One barrier per block while the panels are
  // ahead: the two column-block buffers alternate, and a buffer's next writer has passed the barrier of the block in between.
  v4d lt = col_block(0);
  trailing(0, lt);
  drain_vm();  // (all of X has been used; the fragments of the diagonal tile are next; the panels' DMA has landed)
#pragma unroll
  for (int kc = 0; kc < QB; ++kc) {
    stamp(tr ? tr + 8 + kc : nullptr);
    if (kc == QB - 2 && tid == 0) ctl[5] = wait_flag(a, a.chain_dep[2 * j]) ? 1u : 0u;  // the diagonal tile (normally long up)
    __syncthreads();  // column block kc is in LDS
    if (kc == QB - 2) {
      if (ctl[5] == 0) return false;
#pragma unroll
      for (int m = 0; m < NF; ++m)
        if (live[m]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {   // (buffer loads: 32-bit offsets instead of a 64-bit address per load)
            const int voff = ((16 * fbi[m] + lq + 4 * r) * ld32 + 16 * fbj[m] + lr) * 8;
            pin[m][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rA, voff, tile_dd, 16));
          }
        }
    }
    const bool ahead = kc + 1 < QB && (uint32_t)(kc + 1) < staged;
    if (ahead) lt = col_block(kc + 1);
    fragments(kc);
    stamp(tr ? tr + 16 + kc : nullptr);
    if (kc + 1 < QB) {
      if (!ahead) {  // the next panel was not out when this workgroup last looked: wait for it, stage what is out now
        if (tid == 0) ctl[4] = wait_inputs(kc + 1, false);
        __syncthreads();
        avail = ctl[4];
        if (avail == 0) return false;
        stage_dma((int)staged, (int)avail);
        staged = avail;
        drain_vm();
        __syncthreads();
        lt = col_block(kc + 1);
      }
      trailing(kc + 1, lt);
    }
  }
  drain_vm();  // (this wave's stores of L(j,j-1))
  __syncthreads();
  if (tid == 0) st_flag(a.flags + (uint32_t)(a.ntasks + a.NB) + (uint32_t)(j - 1), 1u);  // L(j,j-1) is out
#pragma unroll
  for (int m = 0; m < NF; ++m) {
    if (!live[m]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * fbi[m] + lq + 4 * r, col = 16 * fbj[m] + lr;
      S[row * QS + col] = (col <= row) ? pin[m][r] - sacc[m][r] : 0.0;
    }
  }
  drain_vm();
  __syncthreads();
  return true;
}

__device__ __attribute__((noinline)) void run_duo(const DagArgs& a_mem, uint32_t c) {
  const DagArgs a = dag_uniform_copy(a_mem);
  DAG_LDS_DECL;
  char* const lds = dag_lds;
  double* const S = (double*)lds;
  WorkItem* const items = (WorkItem*)(S + QN * QS);
  const int tid = threadIdx.x;
  c = uni(c);
  leaf_build_items(items, tid);
  if (tid == 0) *(lds_sync_t*)(lds_char*)(dag_lds + CTL_OFF + 32) = 0;
  __syncthreads();
  const uint32_t WD = (uint32_t)a.ntasks;
#pragma unroll 1
  for (int j = (int)c; j < a.NB; j += 2) {
    unsigned long long* const tr = (a.trace && tid == 0) ? a.trace + CT * j : nullptr;
    if (j == 0) {
      stamp(tr);
      stamp(tr ? tr + 1 : nullptr);
      chain_diag(a, 0, NONE);  // (tile (0,0) as the assembly kernel left it)
    } else if (!duo_helper(a, j)) {
      return;
    }
    stamp(tr ? tr + 2 : nullptr);
    const int64_t off = (int64_t)j * TILE;
    leaf_core<true, NoLeafHook, true>(S, items, a.Lp, a.ld, off, a.info, (lds_sync_t*)(lds_char*)(dag_lds + CTL_OFF + 32),
                                      a.Wp + off * a.ld + off, NoLeafHook(), duo_panel_flags(a) + DUO_PF * j);
    drain_vm();
    __syncthreads();
    if (tid == 0) st_flag(a.flags + WD + (uint32_t)j, 1u);
    stamp(tr ? tr + 3 : nullptr);
  }
}

// One launch, B >= 1 matrices (DagArgs::B; the factor-only plan of the hyper-parameter fit: tgp_nlml_trial_batch): the
// first B workgroups to arrive are the chains of members 0 .. B - 1, everybody else draws from ONE list of
// (member << 24 | task) entries -- the members' dispatch orders interleaved (dag_merge_order), a topological order of
// every member's graph, so the in-order dispatcher stays deadlock-free.  A member that breaks down (not PD) poisons only
// its own tiles.  B <= 1 is the single-matrix `update`, unchanged.
__global__ __launch_bounds__(512) void dag_update_kernel(DagArgs a) {
  DAG_LDS_DECL;
  char* const lds = dag_lds;
  volatile uint32_t* const ctl = (volatile uint32_t*)(lds + CTL_OFF);
  const int tid = threadIdx.x;
  if (tid == 0) ctl[0] = atomicAdd(a.ctrl + C_TICKET, 1u);  // arrival ticket: the first resident workgroups are the chains
  __syncthreads();
  const uint32_t role = ctl[0];
  __syncthreads();
  const uint32_t nB = a.B > 1 ? (uint32_t)a.B : 1u;
  if (a.duo && role < 2u) {
    run_duo(a, role);
    return;
  }
  if (role < nB) {
    if (nB == 1) {
      run_chain(a);
    } else {
      const DagArgs m = dag_member(a, role);
      run_chain(m);
    }
    return;
  }
  const void* const kernarg = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
  const DagArgs wa = dag_args(kernarg);   // the dispatcher's own copy, in scalar registers
  const uint32_t total = (uint32_t)wa.ntasks * nB;
#pragma unroll 1
  for (;;) {
    if (tid == 0) {  // the next entry of the list; wait for the flags of what was drawn
      const uint32_t pos = atomicAdd(wa.ctrl + C_HEAD, 1u);
      uint32_t got = TASK_DONE, mb = 0;
      if (pos < total) {
        const uint32_t entry = wa.topo[pos];
        mb = entry >> 24;
        got = entry & 0xffffffu;
        if (wa.trace) wa.trace[CT * wa.NB + 4 * (size_t)got] = wall_clock64();  // development aid: drawn (before the wait)
        const uint32_t* const mflags = wa.flags + (size_t)mb * wa.flags_stride;
        bool ok = true;
        for (int d = 0; d < 3; ++d) ok = ok && wait_flag_at(mflags, wa.ctrl, wa.tasks[got].dep[d]);
        if (!(wa.tasks[got].flags & DAG_SIB)) ok = ok && wait_flag_at(mflags, wa.ctrl, wa.tasks[got].dep3);
        if (!ok) got = TASK_ERR;
      }
      ctl[0] = got;
      ctl[2] = mb;
    }
    __syncthreads();
    const uint32_t idx = uni(ctl[0]), mb = uni(ctl[2]);
    __syncthreads();
    if (idx >= (uint32_t)wa.ntasks) return;  // TASK_DONE / TASK_ERR
    unsigned long long* const tr = (wa.trace && tid == 0) ? wa.trace + CT * wa.NB + 4 * (size_t)idx : nullptr;
    stamp(tr ? tr + 1 : nullptr);
    if (tid == 0) {  // sanity: a task starts exactly once
      const uint32_t old = atomicAdd(dag_cnt(wa) + (size_t)mb * (uint32_t)wa.ntasks + idx, 1u);
      if (old != 0u) {
        st_flag(wa.ctrl + C_ERR, 3u);
        st_flag(wa.ctrl + C_ERRINFO, idx);
      }
    }
    if (wa.trace && tid == 0) {  // development aid: a task must never start before its producers' flags are up
      for (int d = 0; d < ((wa.tasks[idx].flags & DAG_SIB) ? 3 : 4); ++d) {
        const uint32_t dep = d < 3 ? wa.tasks[idx].dep[d] : wa.tasks[idx].dep3;
        if (dep != NONE && ld_flag(wa.flags + dep) == 0) {
          st_flag(wa.ctrl + C_ERR, 2u);
          st_flag(wa.ctrl + C_ERRINFO, idx);
        }
      }
    }
    if (uni(wa.tasks[idx].flags) & DAG_HALF) run_task<true>(kernarg, mb, idx);
    else run_task<false>(kernarg, mb, idx);
    stamp(tr ? tr + 2 : nullptr);
    if (tr) tr[3] = blockIdx.x;
  }
}

// ---- host: the task list ------------------------------------------------------------------------------------------
struct HostTask {
  DagTask t{};
  std::vector<int> deps;   // producers (host task indices; chain steps are negative codes resolved below)
  int sib = -1;            // DAG_SIB: the lower-half sibling (host index)
  double ready = 0, need = 0;
  bool urgent = false;     // the last burst of a tile, T, E: the per-row chains
};

std::vector<std::pair<int, int>> bursts(int lo, int hi, int burst) {  // long bursts first, the LAST (urgent) ones short
  std::vector<std::pair<int, int>> out;
  int k = lo;
  while (k < hi) {
    const int left = hi - k;
    int n;
    if (left <= 2) n = 1;
    else if (left <= burst + 1) n = std::max(1, left - 2);
    // (round 6: never burst, 1, 1 at the end of a tile with the short bursts of the chain-bound sizes -- the long burst cannot start
    // before its last column is out, three steps before the tile is due, and 78 + 24 + 14 us of dependent tasks do not fit
    // three steps of < 50 us: every fourth step of the chain waited 17 us for its tile (tools/dag_exec_sim.py))
    else if (burst == 4 && left == burst + 2) n = burst - 1;
    else n = burst;
    out.emplace_back(k, k + n);
    k += n;
  }
  return out;
}

}  // namespace

// Build the static list for NB = Npad / 128 block rows: tasks, their dispatch order (a topological order: the start
// order of a simulated launch, see below), dependencies as flag ids.
// flag ids: task n -> n;  chain: W_jj / L_jj ready -> ntasks + j;  L(j+1,j) ready -> ntasks + NB + j.
// `with_inverse` false: the factor only (no X / E tasks: W keeps its diagonal tiles W_jj, which the T tasks need).
// `split_critical` (round 6, the single full update at the chain-bound sizes): between the leaf of step j - 1 and the chain's
// product L(j+1,j) = P(j+1,j) W_jj^T lie two DEPENDENT single products on the workers, T(j+1,j-1) = P(j+1,j-1) W^T and the
// last burst of tile (j+1,j), P(j+1,j) -= L(j+1,j-1) L(j,j-1)^T -- 24 us each plus flag latencies, 53 us after the leaf's flag,
// the second 54-us loop beside the chain (profiles/r05_update_breakdown.txt).  Both are split into two half-tile tasks (rows
// 0 .. 63 / 64 .. 127 of the output: DAG_HALF, DAG_HI): a half of the second needs only ITS half of the first, every other consumer
// of L(i,i-2) waits for both halves (a fourth dependency slot), the chain for both halves of P(j+1,j) (chain_dep[2 NB + j]).
// Every element is still the same sum in the same order: bit-identical to the unsplit plan.
void dag_build(int NB, int64_t ld, std::vector<DagTask>& out_tasks, std::vector<uint32_t>& chain_dep, int& n_urgent,
               std::vector<uint32_t>* topo_out, int workers, bool with_inverse, int batch, int batch_workers,
               std::vector<uint32_t>* batch_out, bool split_critical, bool duo) {
  // k tiles per product: longer bursts amortise a task's fixed ~8 us (N = 8192 is throughput-bound: 8.0 -> 7.65 ms with
  // 8), shorter ones keep the scheduling fine where the chain is the bound (N = 4096: 1.91 ms with 4, 2.08 with 8)
  static const int burst_env = getenv("TGP_DAG_BURST") ? atoi(getenv("TGP_DAG_BURST")) : 0;  // development aid
  // (N = 12288: 24.7 -> 23.7 ms with 16).  The factor-only plan (trial evaluations) runs batched, i.e. throughput-bound
  // at every size: 8 from the start (a task's fixed 8 us is 17 % of a 4-tile burst; one plan for the single and the
  // batched form, so their values stay equal bit for bit)
  const int BURST = burst_env > 0 ? burst_env : (NB >= 88 ? 16 : ((NB >= 48 || !with_inverse) ? 8 : 4));
  constexpr int CH_WD = -1000000, CH_LSUB = -2000000;  // chain producers: CH_WD - j, CH_LSUB - j
  std::vector<HostTask> ts;
  std::vector<int> lastG((size_t)NB * NB, -1), Tid((size_t)NB * NB, -1), Eid((size_t)NB * NB, -1);
  std::vector<int> lastG2((size_t)NB * NB, -1), Tid2((size_t)NB * NB, -1);   // the second (upper) half where a task is split
  auto off = [&](int i, int j) { return (uint32_t)((int64_t)i * TILE * ld + (int64_t)j * TILE); };
  // `duo` plans split EVERY T(i,j) and the last product of EVERY tile below the diagonal (not only the pair next to the chain): per
  // column every row i runs T(i,k) -> last product of (i,k+1) -> T(i,k+1) ..., two dependent tasks that as whole tiles take 22 + 24 us
  // + flags -- no chain step shorter than that can be kept up with (tools/dag_exec_sim.py); as halves they take 31.  The upper half
  // waits for its sibling before it publishes (DAG_SIB), so that a whole-tile consumer still needs ONE flag per operand.
  const bool split_all = split_critical && duo;
  // producer(s) of tile L(i,k), i > k, onto a dependency list.  half < 0: the whole tile (both halves of a split T);
  // half 0 / 1: only rows 0 .. 63 / 64 .. 127 (a half task reading its own rows of the A operand)
  auto add_Lprod = [&](std::vector<int>& deps, int i, int k, int half) {
    if (i == k + 1) {
      deps.push_back(CH_LSUB - k);
      return;
    }
    const int lo = Tid[(size_t)i * NB + k], hi = Tid2[(size_t)i * NB + k];
    if (hi < 0) deps.push_back(lo);
    else if (half < 0) {
      if (!split_all) deps.push_back(lo);   // (split_all: the upper half's flag stands for both, DAG_SIB)
      deps.push_back(hi);
    } else deps.push_back(half == 0 ? lo : hi);
  };
  for (int j = 0; j < NB; ++j)
    for (int i = j; i < NB; ++i) {
      const int hi = i == j ? j - 1 : j;  // the chain adds column j - 1 to the diagonal tile itself
      int prev = -1, prev2 = -1;
      for (auto [k0, k1] : bursts(0, std::max(hi, 0), BURST)) {
        HostTask h;
        h.t.a_mat = DAG_MAT_L; h.t.a_off = off(i, k0);
        h.t.b_mat = DAG_MAT_L; h.t.b_off = off(j, k0);
        h.t.c_mat = h.t.o_mat = DAG_MAT_A; h.t.c_off = h.t.o_off = off(i, j);
        h.t.nk = (uint32_t)(k1 - k0);
        h.t.flags = DAG_BETA | DAG_NEG;
        h.ready = k1 - 1;
        h.need = i == j ? j : (i == j + 1 ? j - 0.5 : j);
        h.urgent = k1 == hi;
        // the last burst of tile (j+1, j): what the chain's next product waits for -- two half-tile tasks in the split plan
        // (a single product: bursts() ends every tile on one)
        const bool split = split_critical && k1 == hi && k1 - k0 == 1 && (i == j + 1 || (split_all && i > j));
        int id_lo = -1, id_hi = -1;
        for (int half = 0; half < (split ? 2 : 1); ++half) {
          HostTask g = h;
          if (split) g.t.flags |= DAG_HALF | (half ? DAG_HI : 0);
          add_Lprod(g.deps, i, k1 - 1, split ? half : -1);
          if (i != j) add_Lprod(g.deps, j, k1 - 1, -1);
          if (prev >= 0) g.deps.push_back(prev);   // (the burst before: a whole-tile task -- only a tile's LAST burst is split)
          if (split_all && half == 1) {
            g.t.flags |= DAG_SIB;
            g.sib = id_lo;
          }
          (half == 0 ? id_lo : id_hi) = (int)ts.size();
          ts.push_back(g);
        }
        prev = id_lo;
        prev2 = id_hi;
      }
      lastG[(size_t)i * NB + j] = prev;
      lastG2[(size_t)i * NB + j] = prev2;
      if (i >= j + 2) {  // L(i,j) = P(i,j) W_jj^T
        HostTask h;
        h.t.a_mat = DAG_MAT_A; h.t.a_off = off(i, j);
        h.t.b_mat = DAG_MAT_W; h.t.b_off = off(j, j);
        h.t.c_mat = h.t.o_mat = DAG_MAT_L; h.t.c_off = h.t.o_off = off(i, j);
        h.t.nk = 1;
        h.t.flags = 0;
        h.deps.push_back(CH_WD - j);
        h.ready = j;
        h.need = j + 1;
        h.urgent = true;
        const bool split = split_critical && (i == j + 2 || split_all);   // T(i, i-2): the first of the two dependent products
        Tid[(size_t)i * NB + j] = (int)ts.size();
        if (split) {
          HostTask lo = h, up = h;
          lo.t.flags |= DAG_HALF;
          up.t.flags |= DAG_HALF | DAG_HI;
          if (prev >= 0) lo.deps.push_back(prev);
          if (prev >= 0) up.deps.push_back(prev2 >= 0 ? prev2 : prev);   // (a half of a split last product: its own rows)
          if (prev2 >= 0 && !split_all) up.deps.push_back(prev);
          if (split_all) {
            up.t.flags |= DAG_SIB;
            up.sib = (int)ts.size();
          }
          ts.push_back(lo);
          Tid2[(size_t)i * NB + j] = (int)ts.size();
          ts.push_back(up);
        } else {
          if (prev >= 0) h.deps.push_back(prev);
          if (prev2 >= 0) h.deps.push_back(prev2);
          ts.push_back(h);
        }
      }
    }
  for (int i = 1; with_inverse && i < NB; ++i)
    for (int c = 0; c < i; ++c) {
      int prev = -1;
      for (auto [k0, k1] : bursts(c, i, BURST)) {  // V(i,c) += sum_k L(i,k) W(k,c),  W(c,c) = W_cc
        HostTask h;
        h.t.a_mat = DAG_MAT_L; h.t.a_off = off(i, k0);
        h.t.b_mat = DAG_MAT_W; h.t.b_off = off(k0, c);
        h.t.c_mat = h.t.o_mat = DAG_MAT_W; h.t.c_off = h.t.o_off = off(i, c);
        h.t.nk = (uint32_t)(k1 - k0);
        h.t.flags = DAG_NN | (prev >= 0 ? DAG_BETA : 0);
        add_Lprod(h.deps, i, k1 - 1, -1);
        h.deps.push_back(k1 - 1 == c ? CH_WD - c : Eid[(size_t)(k1 - 1) * NB + c]);
        if (prev >= 0) h.deps.push_back(prev);
        h.ready = k1 - 1 + 0.5;
        h.need = i;
        h.urgent = k1 == i;
        prev = (int)ts.size();
        ts.push_back(h);
      }
      HostTask h;  // W(i,c) = -W_ii V(i,c)
      h.t.a_mat = DAG_MAT_W; h.t.a_off = off(i, i);
      h.t.b_mat = DAG_MAT_W; h.t.b_off = off(i, c);
      h.t.c_mat = h.t.o_mat = DAG_MAT_W; h.t.c_off = h.t.o_off = off(i, c);
      h.t.nk = 1;
      h.t.flags = DAG_NN | DAG_NEG;
      h.deps.push_back(CH_WD - i);
      h.deps.push_back(prev);
      h.ready = i;
      h.need = i + 1;
      h.urgent = true;
      Eid[(size_t)i * NB + c] = (int)ts.size();
      ts.push_back(h);
    }
  // chain steps as nodes of the same graph: A(j) = leaf (sets W_jj), B(j) = L(j+1,j)
  // (`duo`, the two-workgroup chain: B(j) = the helper's tail after the leaf's last panel, and a third node C(j) = what the helper
  // needs at least between the arrival of P(j+1,j) and the end of its eight column blocks; only the simulation sees it)
  const int nb = (int)ts.size();
  auto chainA = [&](int j) { return nb + 2 * j; };
  auto chainB = [&](int j) { return nb + 2 * j + 1; };
  auto chainC = [&](int j) { return nb + 2 * NB + j; };
  const int total = nb + 3 * NB;
  std::vector<std::vector<int>> deps(total);
  auto resolve = [&](int d) { return d <= CH_LSUB ? chainB(CH_LSUB - d) : (d <= CH_WD ? chainA(CH_WD - d) : d); };
  for (int n = 0; n < nb; ++n)
    for (int d : ts[n].deps) deps[n].push_back(resolve(d));
  chain_dep.assign((size_t)3 * NB, NONE);
  std::vector<int> chain_dep_host((size_t)3 * NB, -1);
  for (int j = 0; j < NB; ++j) {
    if (j > 0) deps[chainA(j)].push_back(chainB(j - 1));
    if (lastG[(size_t)j * NB + j] >= 0) {
      deps[chainA(j)].push_back(lastG[(size_t)j * NB + j]);
      chain_dep_host[2 * j] = lastG[(size_t)j * NB + j];
    }
    deps[chainB(j)].push_back(chainA(j));
    if (duo) deps[chainB(j)].push_back(chainC(j));
    const int bulk_in = duo ? chainC(j) : chainB(j);
    if (j + 1 < NB && lastG[(size_t)(j + 1) * NB + j] >= 0) {
      deps[bulk_in].push_back(lastG[(size_t)(j + 1) * NB + j]);
      chain_dep_host[2 * j + 1] = lastG[(size_t)(j + 1) * NB + j];
      if (lastG2[(size_t)(j + 1) * NB + j] >= 0) {   // P(j+1,j) finished in two halves: the chain waits for both
        deps[bulk_in].push_back(lastG2[(size_t)(j + 1) * NB + j]);
        chain_dep_host[2 * NB + j] = lastG2[(size_t)(j + 1) * NB + j];
      }
    }
    if (duo && j + 1 < NB && lastG[(size_t)(j + 1) * NB + j + 1] >= 0) deps[chainC(j)].push_back(lastG[(size_t)(j + 1) * NB + j + 1]);
  }
  std::vector<int> indeg(total, 0);
  std::vector<std::vector<int>> users(total);
  for (int n = 0; n < total; ++n)
    for (int d : deps[n]) {
      ++indeg[n];
      users[d].push_back(n);
    }
  // DAG_SIB: an upper half is listed behind its sibling (it waits for the sibling's flag at its END: the sibling must have been
  // drawn by then) -- in the simulation it may start as soon as the sibling has STARTED
  std::vector<int> sib_user(total, -1);
  for (int n = 0; n < nb; ++n)
    if (ts[n].sib >= 0) {
      sib_user[ts[n].sib] = n;
      ++indeg[n];
    }
  // Dispatch order = the start order of a LIST-SCHEDULING SIMULATION of the launch (highest level first): `workers`
  // workgroups draw, whenever one is free, the ready task with the longest remaining path to the end of the
  // factorisation; the chain steps run on their own workgroup.  Durations are the measured ones (tools/dag_trace.py,
  // us): a product 6.5 + 19 per k tile (+ 1.5 until its flag is seen), diag + leaf 44, L(j+1,j) 14.5.  The kernel's
  // workers then draw this list IN ORDER and wait for the flags of what they drew: as long as the model is roughly
  // right a task is drawn about when it becomes ready, and what the chain needs next is never queued behind the pile
  // of first bursts whose results are needed twenty steps later (the earlier order -- by the step a task becomes ready
  // at -- left the chain waiting 50 us every fourth step at N = 4096 and 70 us per step at N = 8192).
  std::vector<double> dur(total), tail(total, 0.0);
  for (int n = 0; n < nb; ++n) dur[n] = 8.0 + ((ts[n].t.flags & DAG_HALF) ? 10.0 : 19.0) * ts[n].t.nk;
  for (int j = 0; j < NB; ++j) {
    // (duo, measured -- profiles/r06_dag_duo_v8.txt: leaf 27; the helper's loads 7 + eight column blocks 24 once P(j+1,j) is there,
    // 2 more to hand S over)
    dur[chainA(j)] = duo ? 27.0 : (j == 0 ? 31.0 : 45.0);
    dur[chainB(j)] = j + 1 < NB ? (duo ? 2.0 : 15.0) : 0.0;
    dur[chainC(j)] = duo && j + 1 < NB ? 31.0 : 0.0;
  }
  std::vector<int> full;  // any topological order of all nodes
  {
    std::vector<int> deg = indeg, stack;
    for (int n = 0; n < total; ++n)
      if (deg[n] == 0) stack.push_back(n);
    while (!stack.empty()) {
      const int n = stack.back();
      stack.pop_back();
      full.push_back(n);
      for (int u : users[n])
        if (--deg[u] == 0) stack.push_back(u);
      if (sib_user[n] >= 0 && --deg[sib_user[n]] == 0) stack.push_back(sib_user[n]);
    }
  }
  for (auto it = full.rbegin(); it != full.rend(); ++it) {
    double t = 0.0;
    for (int u : users[*it]) t = std::max(t, tail[u]);
    tail[*it] = t + dur[*it];
  }
  // earliest start with unlimited workers -> slack of a node against the critical path
  std::vector<double> est(total, 0.0);
  double span = 0.0;
  for (int n : full) {
    for (int d : deps[n]) est[n] = std::max(est[n], est[d] + dur[d]);
    span = std::max(span, est[n] + tail[n]);
  }
  std::vector<int> place(total, -1);  // host index -> position in the device array (urgent list, then bulk list)
  std::vector<int> topo;              // the dispatch order (bulk-side nodes only)
  // The simulation, for `members` copies of the graph sharing `nworkers` workers (members = 1: the single update; > 1: a
  // batched launch, every member with a chain workgroup of its own): -> (member, node) in start order.
  auto simulate = [&](int members, int nworkers, std::vector<std::pair<int, int>>& out) {
    // A task enters the simulation when its ready time is KNOWN, i.e. when all its producers have started (chain steps
    // need no worker: they start when ready).  A free worker takes the candidate with the longest remaining path among
    // the tasks that are ready -- and among the near-critical ones (slack < SLACK_CRIT) that will be within LOOK us:
    // it then waits for the flags, which is exactly what the kernel's worker does with an entry it drew too early.  So
    // the list carries the chain's next operands a little AHEAD of their turn and a worker is already parked on them when
    // their producers finish (without this a critical T or G task waited ~13 us for the next worker to come free,
    // twice per step early in the factorisation).  Every producer of a task is still listed before it.
    constexpr double LOOK = 12.0, SLACK_CRIT = 40.0;
    typedef std::pair<double, int> Ev;
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> cand_crit, cand_norm;  // (ready time, id)
    std::priority_queue<double, std::vector<double>, std::greater<double>> busy;      // workers' free times
    std::priority_queue<Ev> avail;                                                    // (tail, -id)
    const int Bm = members;                     // id = node * Bm + member (members = 1: id = node)
    std::vector<int> pend((size_t)total * Bm);  // producers not yet started
    for (int n = 0; n < total; ++n)
      for (int m = 0; m < Bm; ++m) pend[(size_t)n * Bm + m] = indeg[n];
    std::vector<double> ready_at((size_t)total * Bm, 0.0);
    int free_workers = std::max(1, nworkers);
    double now = 0.0;
    std::vector<std::pair<int, double>> work;  // (id, finish time) whose users are to be told
    auto known = [&](int id) {
      const int n = id / Bm;
      if (n >= nb) work.emplace_back(id, ready_at[id] + dur[n]);
      else (span - est[n] - tail[n] < SLACK_CRIT ? cand_crit : cand_norm).push({ready_at[id], id});
    };
    auto started = [&](int id0, double f0) {
      work.emplace_back(id0, f0);
      while (!work.empty()) {
        const auto [id, f] = work.back();
        work.pop_back();
        const int n = id / Bm, m = id % Bm;
        for (int u : users[n]) {
          const int uid = u * Bm + m;
          ready_at[uid] = std::max(ready_at[uid], f);
          if (--pend[uid] == 0) known(uid);
        }
        if (sib_user[n] >= 0) {  // (the sibling: from this task's START)
          const int uid = sib_user[n] * Bm + m;
          ready_at[uid] = std::max(ready_at[uid], f - dur[n]);
          if (--pend[uid] == 0) known(uid);
        }
      }
    };
    for (int n = 0; n < total; ++n)
      if (indeg[n] == 0)
        for (int m = 0; m < Bm; ++m) {
          known(n * Bm + m);
          while (!work.empty()) {  // (a chain step without producers: step 0's leaf)
            const auto [c, f] = work.back();
            work.pop_back();
            started(c, f);
          }
        }
    for (;;) {
      while (!cand_crit.empty() && cand_crit.top().first <= now + LOOK) {
        avail.push({tail[cand_crit.top().second / Bm], -cand_crit.top().second});
        cand_crit.pop();
      }
      while (!cand_norm.empty() && cand_norm.top().first <= now) {
        avail.push({tail[cand_norm.top().second / Bm], -cand_norm.top().second});
        cand_norm.pop();
      }
      if (free_workers > 0 && !avail.empty()) {
        const int id = -avail.top().second;
        avail.pop();
        const double f = std::max(now, ready_at[id]) + dur[id / Bm];
        out.emplace_back(id % Bm, id / Bm);
        busy.push(f);
        --free_workers;
        started(id, f);
        continue;
      }
      double next = 1e300;
      if (!busy.empty()) next = std::min(next, busy.top());
      if (free_workers > 0) {
        if (!cand_crit.empty()) next = std::min(next, cand_crit.top().first - LOOK);
        if (!cand_norm.empty()) next = std::min(next, cand_norm.top().first);
      }
      if (next >= 1e300) break;
      now = std::max(now, next);
      while (!busy.empty() && busy.top() <= now) {
        busy.pop();
        ++free_workers;
      }
    }
  };
  {
    std::vector<std::pair<int, int>> one;
    simulate(1, workers, one);
    for (const auto& e : one) topo.push_back(e.second);
  }
  std::vector<int> order;
  for (int pass = 0; pass < 2; ++pass) {
    for (int n : topo)
      if (ts[n].urgent == (pass == 0)) {
        place[n] = (int)order.size();
        order.push_back(n);
      }
    if (pass == 0) n_urgent = (int)order.size();
  }
  if (topo_out) {
    topo_out->clear();
    for (int n : topo) topo_out->push_back((uint32_t)place[n]);
  }
  if (batch_out && batch > 1) {
    // the dispatch list of a batched launch: `batch` members simulated TOGETHER on the workers they share -- a free
    // worker takes the most critical ready task of ANY member, so the members fill each other's chain-bound gaps
    // (round 4's first form interleaved B copies of one member's order entry by entry: workers 81 % busy)
    std::vector<std::pair<int, int>> all;
    simulate(batch, std::max(1, batch_workers), all);
    batch_out->clear();
    for (const auto& e : all) batch_out->push_back(((uint32_t)e.first << 24) | (uint32_t)place[e.second]);
  }
  const uint32_t WD = (uint32_t)nb, LSUB = (uint32_t)(nb + NB);
  auto flag_of = [&](int d) -> uint32_t {
    if (d <= CH_LSUB) return LSUB + (uint32_t)(CH_LSUB - d);
    if (d <= CH_WD) return WD + (uint32_t)(CH_WD - d);
    return (uint32_t)place[d];
  };
  out_tasks.resize(nb);
  for (int p = 0; p < nb; ++p) {
    const HostTask& h = ts[order[p]];
    DagTask t = h.t;
    t.dep[0] = t.dep[1] = t.dep[2] = t.dep3 = NONE;
    if (h.deps.size() > (h.sib >= 0 ? 3u : 4u)) abort();   // (the plan never needs more: see split_critical above)
    for (size_t d = 0; d < h.deps.size(); ++d) (d < 3 ? t.dep[d] : t.dep3) = flag_of(h.deps[d]);
    if (h.sib >= 0) t.dep3 = (uint32_t)place[h.sib];
    t.set = (uint32_t)p;
    out_tasks[p] = t;
  }
  for (int s = 0; s < 3 * NB; ++s)
    if (chain_dep_host[s] >= 0) chain_dep[s] = (uint32_t)place[chain_dep_host[s]];
}

size_t dag_lds_bytes() { return DAG_LDS; }

namespace {
// z = L^-1 r by block forward substitution over the 128-blocks, given the diagonal inverses W_jj (what a factor-only
// launch leaves in W): workgroup i owns block row i, adds L(i,k) z_k to its sum as the z_k appear (one flag per block,
// same hand-over as the tile tasks: write-through store, drain, barrier, flag) and then publishes z_i = W_ii (r_i - sum).
// Workgroups are dispatched in index order and wait only for lower indices: no deadlock whatever the residency.
// A wait that times out poisons z_i with NaN (the caller sees a NaN likelihood).
// Batched (gridDim.y members: matrices mat_stride apart, vectors ld apart, flags gridDim.x words apart): workgroups are
// dispatched x fastest, so member y's block i still waits only for lower linear indices.
__global__ __launch_bounds__(256) void block_trsv_kernel(const double* __restrict__ L, const double* __restrict__ W,
                                                         int64_t ld, const double* __restrict__ r,
                                                         double* z, uint32_t* flags, int64_t mat_stride) {
  __shared__ double zk[TILE], part[2][TILE], t[TILE];
  __shared__ uint32_t ok;
  L += (int64_t)blockIdx.y * mat_stride;
  W += (int64_t)blockIdx.y * mat_stride;
  r += (int64_t)blockIdx.y * ld;
  z += (int64_t)blockIdx.y * ld;
  flags += (size_t)blockIdx.y * gridDim.x;
  const int i = blockIdx.x, tid = threadIdx.x, row = tid & (TILE - 1), half = tid >> 7;
  const double* const Lrow = L + ((int64_t)i * TILE + row) * ld + half * (TILE / 2);
  double s = 0.0;
  bool good = true;
  for (int k = 0; k < i; ++k) {
    if (tid == 0) {
      unsigned spins = 0;
      while (ld_flag(flags + k) == 0 && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(4);
      ok = spins < SPIN_LIMIT ? 1u : 0u;
    }
    __syncthreads();
    good = good && ok != 0;
    if (tid < TILE) zk[tid] = ld8_sc1(z + (int64_t)k * TILE + tid);
    __syncthreads();
    const double* const lp = Lrow + (int64_t)k * TILE;
#pragma unroll 8
    for (int c = 0; c < TILE / 2; ++c) s = fma(lp[c], zk[half * (TILE / 2) + c], s);
  }
  part[half][row] = s;
  __syncthreads();
  if (tid < TILE) t[tid] = r[(int64_t)i * TILE + tid] - (part[0][tid] + part[1][tid]);
  __syncthreads();
  const double* const wp = W + ((int64_t)i * TILE + row) * ld + (int64_t)i * TILE + half * (TILE / 2);
  double acc = 0.0;  // (W_ii carries explicit zeros right of its diagonal)
#pragma unroll 8
  for (int c = 0; c < TILE / 2; ++c) acc = fma(wp[c], t[half * (TILE / 2) + c], acc);
  __syncthreads();
  part[half][row] = acc;
  __syncthreads();
  if (tid < TILE) {
    const double v = good ? part[0][tid] + part[1][tid] : __builtin_nan("");
    __hip_atomic_store((gdouble*)(z + (int64_t)i * TILE + tid), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  drain_vm();
  __syncthreads();
  if (tid == 0) st_flag(flags + i, 1u);
}
}  // namespace

void launch_block_trsv(hipStream_t s, const double* L, const double* W, int64_t ld, int NB, const double* r, double* z,
                       uint32_t* flags, int B, int64_t mat_stride) {
  hipLaunchKernelGGL(block_trsv_kernel, dim3((unsigned)NB, (unsigned)std::max(1, B)), dim3(256), 0, s, L, W, ld, r, z, flags,
                     mat_stride);
}

// Dispatch list of a batched launch: B copies of one member's order, interleaved entry by entry -- the members are the
// same graph with the same simulated start times, so this IS the merge by start time of a simulation in which every
// member owns 1 / B of the workers (build the member's order with dag_build(..., workers / B, ...)).  Every member's
// entries keep their relative order: a topological order of every member's graph.
void dag_merge_order(const std::vector<uint32_t>& member_order, int B, std::vector<uint32_t>& merged) {
  merged.clear();
  merged.reserve(member_order.size() * (size_t)B);
  for (uint32_t t : member_order)
    for (int b = 0; b < B; ++b) merged.push_back(((uint32_t)b << 24) | t);
}

hipError_t launch_dag_update(hipStream_t s, const DagArgs& a, int grid) {
  hipError_t e = hipFuncSetAttribute((const void*)dag_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DAG_LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dag_update_kernel, dim3((unsigned)grid), dim3(512), DAG_LDS, s, a);
  return hipGetLastError();
}

}  // namespace tgp
