// Stream-K form of the f64 MFMA product for the LARGE nodes of the factor recursion (node size >= 1024).
//
// The products of a node multiply by a triangular factor, so the k range of an output tile depends on the tile
// (`tri`, see tgp_kernels_linalg.hip), and a node of size 1024 / 2048 has only 64 / 256 tiles of 128 x 128 -- the tile
// size that is not L2-bandwidth bound -- with k ranges from 128 to the full size: one tile per workgroup leaves the
// makespan at the longest tile (the 64 x 64-tile kernel's way out is four times the tiles at 8 flop per operand byte:
// 50 TFLOP/s measured on these products).  Here the WORK, not the tiles, is divided: the k ranges of all live tiles,
// laid end to end in a fixed tile order, are cut into `gridDim.x` equal runs of 64-deep units, one run per workgroup
// (<= one workgroup per CU: all resident).  A workgroup whose run STARTS inside a tile leaves that first partial
// accumulator in a scratch slot and raises a flag before it goes on; the workgroup that holds the START of a tile --
// it reaches it as the last segment of its run -- adds the partials of its successors in workgroup order (a fixed
// order, so results are reproducible bit for bit, unlike an atomic-add fixup) and writes C.  Nobody waits before
// the end of its own run, and the partials waited for were the first thing their producers did: no chain of waits
// (the mirrored arrangement -- owners at the tile ends -- serialised the whole grid: 1.4 ms per launch).  The wait
// is bounded anyway (a lost flag sets `*info` instead of hanging the GPU).
//
// Reference call sites replaced: the triangular solves / matrix products inside tf.linalg.cholesky and
// tf.linalg.triangular_solve of gpflow's GPR posterior (trieste/models/gpflow/models.py:171-186 -> interface.py:108-112).
#include "tgp_dev.hpp"
#include "tgp_internal.hpp"

namespace tgp {
namespace {

constexpr int SKB = 128, SKK = 16, SKLD = SKB + 16;  // tile, k-step, LDS row stride
constexpr int SKU = 64;                               // work unit: 64-deep slice of one tile's k range

struct SkProblem {
  int m, n, k;
  double alpha, beta;
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int lower_only, tri;
  double* scratch;      // [gridDim.x][32][512]
  unsigned* flags;      // [gridDim.x], holds the epoch of the last launch that left a partial there
  unsigned epoch;
  int* info;
};

// units of the tiles of "major" index j (the index the k range depends on) -- see `tri` in tgp_kernels_linalg.hip
__device__ __forceinline__ int sk_len(const SkProblem& p, int j) {
  const int full = p.k / SKU, per = SKB / SKU;
  switch (p.tri) {
    case 1: case 3: return min(full, (j + 1) * per);   // k < (j + 1) 128
    case 2: case 5: return full - min(full, j * per);  // k >= j 128
    default: return full;
  }
}
__device__ __forceinline__ int sk_klo(const SkProblem& p, int j) {
  return (p.tri == 2 || p.tri == 5) ? min(p.k, j * SKB) : 0;
}

struct SkTile {
  int tm, tn, first_unit, len, klo;
};
// the tile that holds unit u.  Tile order: lower_only -- the tiles tn <= tm row by row (all of equal length);
// tri 1, 2 -- column-major (major = tn); tri 3, 5 and unpruned -- row-major (major = tm).
__device__ __forceinline__ SkTile sk_locate(const SkProblem& p, int u) {
  const int ntm = p.m / SKB, ntn = p.n / SKB;
  SkTile t;
  if (p.lower_only) {
    t.len = p.k / SKU;
    const int id = u / t.len;
    int r = (int)((sqrt(8.0 * id + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= id) ++r;
    while (r * (r + 1) / 2 > id) --r;
    t.tm = r;
    t.tn = id - r * (r + 1) / 2;
    t.first_unit = id * t.len;
    t.klo = 0;
    return t;
  }
  const bool col_major = p.tri == 1 || p.tri == 2;
  const int nmajor = col_major ? ntn : ntm, nminor = col_major ? ntm : ntn;
  int cum = 0, j = 0, len = sk_len(p, 0);
  while (j + 1 < nmajor && cum + nminor * len <= u) {
    cum += nminor * len;
    ++j;
    len = sk_len(p, j);
  }
  const int minor = (u - cum) / len;
  t.len = len;
  t.first_unit = cum + minor * len;
  t.klo = sk_klo(p, j);
  t.tm = col_major ? minor : j;
  t.tn = col_major ? j : minor;
  return t;
}

template <bool TB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_sk_kernel(const SkProblem p, int total_units) {
  __shared__ __attribute__((aligned(16))) double sm[2][2][SKK][SKLD];  // [stage][A|B][k][row/col]
  __shared__ int timed_out;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 2, wn = w & 3;  // 2 x 4 waves, wave tile 64 x 32
  const int g = blockIdx.x, G = gridDim.x;
  const int u0 = (int)((int64_t)g * total_units / G), u1 = (int)((int64_t)(g + 1) * total_units / G);
  const int lr = tid >> 2, lk = (tid & 3) * 4;   // [row][k..k+3] loader (A, and B when TB)
  const int br = tid >> 5, bc = (tid & 31) * 4;  // [k][n..n+3] loader (B when !TB)
  if (tid == 0) timed_out = 0;

  for (int u = u0; u < u1;) {
    const SkTile t = sk_locate(p, u);
    const int tile_end = t.first_unit + t.len, seg_end = min(u1, tile_end);
    const int ka = t.klo + (u - t.first_unit) * SKU, kb = t.klo + (seg_end - t.first_unit) * SKU;
    v4d acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    const double* Ab = p.A + (int64_t)t.tm * SKB * p.lda;
    const double* Bb = TB ? p.B + (int64_t)t.tn * SKB * p.ldb : p.B + (int64_t)t.tn * SKB;
    v2d a0, a1, b0, b1;
    auto fetch = [&](int k0) {
      const double* sa = Ab + (int64_t)lr * p.lda + k0 + lk;
      a0 = *(const v2d*)sa;
      a1 = *(const v2d*)(sa + 2);
      const double* sb = TB ? Bb + (int64_t)lr * p.ldb + k0 + lk : Bb + (int64_t)(k0 + br) * p.ldb + bc;
      b0 = *(const v2d*)sb;
      b1 = *(const v2d*)(sb + 2);
    };
    auto stage = [&](int st) {
      double(*As)[SKLD] = sm[st][0];
      double(*Bs)[SKLD] = sm[st][1];
      const int sc = (lr + 2 * lk) & (SKB - 1);  // column rotation by 8 (k >> 2): see gemm_kernel8
      As[lk + 0][sc] = a0.x; As[lk + 1][sc] = a0.y; As[lk + 2][sc] = a1.x; As[lk + 3][sc] = a1.y;
      if (TB) {
        Bs[lk + 0][sc] = b0.x; Bs[lk + 1][sc] = b0.y; Bs[lk + 2][sc] = b1.x; Bs[lk + 3][sc] = b1.y;
      } else {
        *(v2d*)&Bs[br][bc] = b0;
        *(v2d*)&Bs[br][bc + 2] = b1;
      }
    };
    __syncthreads();  // the previous segment's readers are done with the LDS stages
    fetch(ka);
    stage(0);
    __syncthreads();
    int st = 0;
    for (int k0 = ka; k0 < kb; k0 += SKK) {
      const bool more = k0 + SKK < kb;
      if (more) fetch(k0 + SKK);
#pragma unroll
      for (int k4 = 0; k4 < SKK / 4; ++k4) {
        const int kr = k4 * 4 + (lane >> 4);
        double av[4], bv[2];
#pragma unroll
        for (int f = 0; f < 4; ++f) av[f] = sm[st][0][kr][(wm * 64 + f * 16 + (lane & 15) + 8 * k4) & (SKB - 1)];
#pragma unroll
        for (int f = 0; f < 2; ++f)
          bv[f] = sm[st][1][kr][TB ? (wn * 32 + f * 16 + (lane & 15) + 8 * k4) & (SKB - 1) : wn * 32 + f * 16 + (lane & 15)];
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
          for (int fn = 0; fn < 2; ++fn) acc[fm][fn] = mfma_f64(av[fm], bv[fn], acc[fm][fn]);
      }
      if (more) stage(st ^ 1);
      __syncthreads();
      st ^= 1;
    }

    if (u > t.first_unit) {
      // the run starts inside this tile (so this is its first segment): leave the partial for the tile's owner, the
      // workgroup that holds the tile's start -- at once, before the rest of the run, so that nobody waits for long.
      // (agent-scope relaxed atomics = stores that go through to the coherent level on their own: no cache-wide
      // write-back or invalidate)
      double* slot = p.scratch + (size_t)g * (32 * 512) + tid;
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(slot + ((fm * 2 + fn) * 4 + r) * 512, acc[fm][fn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's stores have been acknowledged
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.flags + g, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (seg_end < tile_end) {
        // owner of a tile that later workgroups finish (this is the last segment of the run): add their partials in
        // workgroup order.  They are the FIRST segments of those runs, raised long ago.
        const int glast = (int)((((int64_t)tile_end) * G - 1) / total_units);  // the workgroup holding unit tile_end - 1
        for (int gp = g + 1; gp <= glast; ++gp) {
          if (tid == 0) {
            long spins = 0;
            while (__hip_atomic_load(p.flags + gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1L << 24)) {  // ~ seconds: never in a correct run
                timed_out = 1;
                break;
              }
            }
          }
          __syncthreads();
          const double* slot = p.scratch + (size_t)gp * (32 * 512) + tid;
          // 16 loads in flight at a time (accumulating straight from the atomic loads serialises them; 32 at once
          // spill the accumulators)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double part[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
              part[e] = __hip_atomic_load(slot + (16 * h + e) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int q = 16 * h + e;
              acc[q >> 3][(q >> 2) & 1][q & 3] += part[e];
            }
          }
        }
        if (timed_out && tid == 0) atomicCAS(p.info, 0, -1);
      }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = (int64_t)t.tm * SKB + wm * 64 + fm * 16 + (lane >> 4) + 4 * r;
            const int64_t col = (int64_t)t.tn * SKB + wn * 32 + fn * 16 + (lane & 15);
            double* dst = p.C + row * p.ldc + col;
            const double v = p.alpha * acc[fm][fn][r];
            *dst = (p.beta == 0.0) ? v : fma(p.beta, *dst, v);
          }
    }
    u = seg_end;
  }
}

}  // namespace

// units of the whole product (host side of sk_len / the tile orders above)
static int64_t sk_total_units(int m, int n, int k, bool lower_only, int tri) {
  const int64_t ntm = m / SKB, ntn = n / SKB, full = k / SKU, per = SKB / SKU;
  if (lower_only) return ntm * (ntm + 1) / 2 * full;
  const bool col_major = tri == 1 || tri == 2;
  const int64_t nmajor = col_major ? ntn : ntm, nminor = col_major ? ntm : ntn;
  int64_t total = 0;
  for (int64_t j = 0; j < nmajor; ++j) {
    int64_t len = full;
    if (tri == 1 || tri == 3) len = std::min(full, (j + 1) * per);
    else if (tri == 2 || tri == 5) len = full - std::min(full, j * per);
    total += nminor * len;
  }
  return total;
}

bool gemm_sk_supported(int m, int n, int k, bool lower_only, int tri) {
  if (m % SKB || n % SKB || k % SKB) return false;
  if (lower_only) return tri == 0 && m == n;
  return tri == 0 || tri == 1 || tri == 2 || tri == 3 || tri == 5;
}

void launch_gemm_sk(hipStream_t s, const StreamKWs& ws, bool tb, int m, int n, int k, double alpha, const double* A,
                    int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, bool lower_only,
                    int tri, int* info) {
  const int64_t total = sk_total_units(m, n, k, lower_only, tri);
  const int grid = (int)std::min<int64_t>(ws.n_wg, total);  // n_wg = 2 per CU: all resident (<= 128 VGPRs, 72 KiB of LDS each)
  SkProblem p{m, n, k, alpha, beta, A, B, C, lda, ldb, ldc, lower_only ? 1 : 0, tri, ws.scratch, ws.flags,
              ++*ws.epoch, info};
  if (tb) hipLaunchKernelGGL(gemm_sk_kernel<true>, dim3((unsigned)grid), dim3(512), 0, s, p, (int)total);
  else hipLaunchKernelGGL(gemm_sk_kernel<false>, dim3((unsigned)grid), dim3(512), 0, s, p, (int)total);
}

}  // namespace tgp
