// Development aid (round 5): what does it cost a CU to ISSUE its operand traffic, and what does that traffic do to waves that
// read LDS and feed the int8 matrix pipe next to it?  The int8 sweep's step trace (profiles/r05_i8_trace_ko.txt) shows 40
// global_load_lds_dwordx4 per step taking ~1350 cycles to issue and a partner wave's LDS-read + MFMA phase slowing 2.2 x
// while they do.  One 512-thread workgroup per CU, everything L2 / MALL resident (every workgroup reads the same 8 MiB).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_vmem.hip -o tools/ubench_vmem && tools/ubench_vmem
// modes: 0 all waves: 4 plain global_load_dwordx4 (1 KiB each) per iteration, three sets in flight
//        1 all waves: 4 global_load_lds_dwordx4 per iteration (m0 saved / restored around each, as the kernels do)
//        2 all waves: compute only (8 ds_read_b128 + 20 v_mfma_i32_32x32x32_i8 per iteration)
//        3 waves 0-3: 10 LDS-DMA per iteration; waves 4-7: compute          (loader | consumer split)
//        4 waves 0-3: 10 plain loads per iteration; waves 4-7: compute
//        5 all waves: 4 plain loads (A operand to registers, used two iterations later) + compute with 8 ds_read + [1 DMA]
//        6 all waves: 5 LDS-DMA then compute with 12 ds_read                 (the shipped kernel's step)
//        7 as 1 without the m0 save / restore (m0 written once per instruction)
//        8 as 6 with the DMA issued between the two MFMA groups
//        9 as 8 with ONE barrier per two iterations (a four-stage ring)
//       10 as 8 with one DMA after every fourth MFMA
//       11 MFMAs only (no LDS reads, no barrier): the floor
//       12 as 8 without the barrier
//       13 as 10 with the second fragment's A operands requested before the first fragment's MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int ITER = 3000, NW = 8;
constexpr size_t BUF = 8u << 20;

__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds16_nom0(const void* g, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
#define PLAIN(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const unsigned char* __restrict__ buf, unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  for (int i = tid; i < 120 * 1024 / 4; i += 512) ((int*)lds)[i] = i * 2654435761u;
  __syncthreads();
  v16i acc[2][4];
  for (int f = 0; f < 2; ++f) for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) acc[f][g][r] = 0;
  v4i ra[3][4];
  for (int s = 0; s < 3; ++s) for (int j = 0; j < 4; ++j) ra[s][j] = (v4i){lane, s, j, 1};
  const bool loader = (MODE == 3 || MODE == 4) && w < 4;
  const bool do_compute = MODE == 2 || MODE == 5 || MODE == 6 || MODE >= 8 || ((MODE == 3 || MODE == 4) && w >= 4);
  const unsigned char* gp = buf + (size_t)w * 4096 + lane * 16;
  size_t goff = 0;
  auto next_ptr = [&]() { const unsigned char* p = gp + goff; goff = (goff + 32768) & (BUF - 1); return p; };
  v4i dfa[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, dfb[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  auto compute = [&](int it, const v4i (&a_reg)[4], bool a_from_reg, bool dma_mid) {
    if (MODE == 10 || MODE == 11 || MODE == 13) {
      const unsigned char* sA = lds + (it % 3) * 40960;
      const unsigned char* sB = sA + 32768;
      v4i bv[4], av[2][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = MODE == 11 ? (v4i){lane, s, it, 1} : *(const v4i*)(sB + s * 2048 + ((w & 1) * 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
      int nd = 0, nm = 0;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        if (MODE != 13 || f == 0) {
#pragma unroll
          for (int ff = f; ff < (MODE == 13 ? 2 : f + 1); ++ff)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              av[ff][s] = MODE == 11 ? (v4i){lane, s, ff, it} : *(const v4i*)(sA + s * 8192 + ((w >> 1) * 64 + ff * 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int s = 0; s <= g; ++s) {
            acc[f][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[f][s], bv[g - s], acc[f][g], 0, 0, 0);
            if (MODE != 11 && (++nm & 3) == 0 && nd < 5) {
              glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + (w * 5 + nd) * 1024));
              ++nd;
            }
          }
      }
      return;
    }
    if (MODE == 14 || MODE == 15) {
      const unsigned char* sA = lds + (it % 3) * 40960;
      const unsigned char* sB = sA + 32768;
      // deferred from the previous iteration: diagonal 3 (MODE 14: 4 MFMAs) or diagonals 2 and 3 (MODE 15: 7) of fragment 1
#pragma unroll
      for (int g = (MODE == 14 ? 3 : 2); g < 4; ++g)
#pragma unroll
        for (int s = 0; s <= g; ++s) acc[1][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(dfa[s], dfb[g - s], acc[1][g], 0, 0, 0);
      v4i bv[4], av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = *(const v4i*)(sB + s * 2048 + ((w & 1) * 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) av[s] = *(const v4i*)(sA + s * 8192 + ((w >> 1) * 64 + (lane & 31)) * 32 + (lane >> 5) * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s <= g; ++s) acc[0][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], bv[g - s], acc[0][g], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 5; ++j) glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + (w * 5 + j) * 1024));
#pragma unroll
      for (int s = 0; s < 4; ++s) av[s] = *(const v4i*)(sA + s * 8192 + ((w >> 1) * 64 + 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
#pragma unroll
      for (int g = 0; g < (MODE == 14 ? 3 : 2); ++g)
#pragma unroll
        for (int s = 0; s <= g; ++s) acc[1][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], bv[g - s], acc[1][g], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) { dfa[s] = av[s]; dfb[s] = bv[s]; }
      return;
    }
    if (MODE == 16 || MODE == 17) {
      // wave grid 8 x 1: one A fragment per plane in REGISTERS (a_reg, landed), two B fragments from LDS; the iteration's
      // vector memory traffic (1 LDS-DMA of the B tile + 4 plain loads of the A operand two tiles ahead) goes out between the
      // two B fragments' MFMAs (16) or before them (17)
      const unsigned char* sB = lds + (it % 3) * 40960 + 32768;
      v4i bv[2][4];
#pragma unroll
      for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[bf][s] = *(const v4i*)(sB + s * 2048 + (bf * 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
      auto traffic = [&]() {
        glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + 32768 + w * 1024));
#pragma unroll
        for (int j = 0; j < 4; ++j) PLAIN(ra[(it + 2) % 3][j], next_ptr());
      };
      if (MODE == 17) traffic();
#pragma unroll
      for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int s = 0; s <= g; ++s) acc[bf][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_reg[s], bv[bf][g - s], acc[bf][g], 0, 0, 0);
        if (MODE == 16 && bf == 0) traffic();
      }
      return;
    }
    const unsigned char* sA = lds + (MODE == 9 ? (it & 3) * 30720 : (it % 3) * 40960);
    const unsigned char* sB = sA + 32768;
    v4i bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[s] = *(const v4i*)(sB + s * 2048 + ((w & 1) * 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      v4i av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        av[s] = (a_from_reg && f == 0) ? a_reg[s]
                : (a_from_reg ? *(const v4i*)(sB + s * 2048 + ((1 - (w & 1)) * 32 + (lane & 31)) * 32 + (lane >> 5) * 16)
                              : *(const v4i*)(sA + s * 8192 + ((w >> 1) * 64 + f * 32 + (lane & 31)) * 32 + (lane >> 5) * 16));
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s <= g; ++s) acc[f][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], bv[g - s], acc[f][g], 0, 0, 0);
      if (dma_mid && f == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + (MODE == 9 ? ((it + 2) & 3) * 30720 : ((it + 2) % 3) * 40960) + (w * 5 + j) * (MODE == 9 ? 768 : 1024)));
      }
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it3 = 0; it3 < ITER; it3 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int it = it3 + u;
      if (MODE == 0 || MODE == 5) {
        if (MODE == 5) glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + 32768 + w * 1024));
#pragma unroll
        for (int j = 0; j < 4; ++j) PLAIN(ra[u][j], next_ptr());
        if (MODE == 0) wait_vm<8>();
        else wait_vm<10>();  // set (u + 1) % 3, loaded two iterations ago, has landed
        asm volatile("" : "+v"(ra[(u + 1) % 3][0]), "+v"(ra[(u + 1) % 3][1]), "+v"(ra[(u + 1) % 3][2]), "+v"(ra[(u + 1) % 3][3]));
      }
      if (MODE == 1 || MODE == 7) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t d = __builtin_amdgcn_readfirstlane(lds0 + (it % 3) * 40960 + (w * 4 + j) * 1024);
          if (MODE == 1) glds16(next_ptr(), d);
          else glds16_nom0(next_ptr(), d);
        }
        wait_vm<8>();
      }
      if (MODE == 6) {
#pragma unroll
        for (int j = 0; j < 5; ++j) glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + (w * 5 + j) * 1024));
      }
      if (loader) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          if (MODE == 3) glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + (w * 10 + j) * 1024));
          else PLAIN(ra[j % 3][j & 3], next_ptr());
        }
        wait_vm<20>();
      }
      if (do_compute) {
        compute(it, ra[(MODE == 16 || MODE == 17) ? u : (u + 1) % 3], MODE == 5, MODE == 8 || MODE == 9 || MODE == 12);
        if (MODE == 6 || MODE == 8 || MODE == 10 || MODE == 12 || MODE == 13 || MODE == 14 || MODE == 15) wait_vm<5>();
        if (MODE == 16 || MODE == 17) {   // everything older than this iteration's five operations has landed: the next A set is usable
          wait_vm<5>();
          asm volatile("" : "+v"(ra[(u + 1) % 3][0]), "+v"(ra[(u + 1) % 3][1]), "+v"(ra[(u + 1) % 3][2]), "+v"(ra[(u + 1) % 3][3]));
        }
        if (MODE == 9 && (it & 1)) wait_vm<5>();
        if (MODE == 2 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 10 || MODE == 13 || MODE == 14 || MODE == 15 || MODE == 16 || MODE == 17 || (MODE == 9 && (it & 1))) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int f = 0; f < 2; ++f) for (int g = 0; g < 4; ++g) s += acc[f][g][0] + acc[f][g][7];
  for (int q = 0; q < 3; ++q) for (int j = 0; j < 4; ++j) s += ra[q][j][0];
  if (s == 0x12345678) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * NW + w] = t1 - t0;
}

// does f64 VALU work of one wave overlap the int8 MFMAs of its SIMD partner?  OV: 1 waves 0-3 MFMA only, 2 waves 4-7 VALU only,
// 3 both; F64: the VALU work is v_fma_f64 (else v_fma_f32); 320 independent-chain FMAs (8 chains) per iteration vs 20 MFMAs
template <int OV, bool F64>
__global__ __launch_bounds__(512, 2) void kov(unsigned long long* out, int* sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  v16i acc[2][4];
  for (int f = 0; f < 2; ++f) for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) acc[f][g][r] = 0;
  double d[8]; float e[8];
  for (int i = 0; i < 8; ++i) { d[i] = 1.0 + lane * 1e-3 + i; e[i] = 1.0f + lane * 1e-3f + i; }
  const v4i av = {lane, 1, 2, 3}, bv = {3, lane, 1, 0};
  const bool do_m = (OV & 1) && w < 4, do_v = (OV & 2) && w >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    if (do_m) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int s = 0; s <= g; ++s) acc[f][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[f][g], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int r = 0; r < 40; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[i]) : "v"(e[(i + 1) & 7]), "v"(e[(i + 2) & 7]));
        }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int f = 0; f < 2; ++f) for (int g = 0; g < 4; ++g) s += acc[f][g][0];
  for (int i = 0; i < 8; ++i) s += (int)d[i] + (int)e[i];
  if (s == 0x12345678) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * NW + w] = t1 - t0;
}
template <int OV, bool F64>
void run_ov(unsigned long long* out, int* sink, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kov<OV, F64><<<256, 512>>>(out, sink);
  hipEventRecord(e0);
  kov<OV, F64><<<256, 512>>>(out, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * NW);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int blk = 0; blk < 256; ++blk) for (int w = 0; w < NW; ++w) (w < 4 ? a : b) += (double)h[blk * NW + w];
  printf("overlap %d %s  %-60s kernel %7.3f ms = %6.0f ns/iter | cycles/iter MFMA waves: %6.0f  VALU waves: %6.0f\n", OV, F64 ? "f64" : "f32",
         what, ms, ms * 1e6 / ITER, a / (256 * 4.0 * ITER), b / (256 * 4.0 * ITER));
  fflush(stdout);
}

// ONE wave per SIMD (256 threads, 512 registers): wave tile 64 x 64 = 16 accumulator fragments, 16 ds_read_b128 + 40 MFMA + 10 DMA per
// iteration.  V: 0 reads up front, DMA after every fourth MFMA; 1 the same with the B-fragment-1 / A-fragment-1 reads issued after the
// first ten MFMAs; 2 as 0 without DMA; 3 as 0, the last eight MFMAs deferred behind the barrier
template <int V>
__global__ __launch_bounds__(256) void k512(const unsigned char* __restrict__ buf, unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  for (int i = tid; i < 120 * 1024 / 4; i += 256) ((int*)lds)[i] = i * 2654435761u;
  __syncthreads();
  v16i acc[2][2][4];
  for (int f = 0; f < 2; ++f) for (int h = 0; h < 2; ++h) for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) acc[f][h][g][r] = 0;
  const unsigned char* gp = buf + (size_t)w * 4096 + lane * 16;
  size_t goff = 0;
  auto next_ptr = [&]() { const unsigned char* p = gp + goff; goff = (goff + 32768) & (BUF - 1); return p; };
  v4i dav[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, dbv[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    const unsigned char* sA = lds + (it % 3) * 40960;
    const unsigned char* sB = sA + 32768;
    if (V == 3) {  // the MFMAs deferred from the previous iteration: the pipe has work while the first reads fly
#pragma unroll
      for (int g = 2; g < 4; ++g)
#pragma unroll
        for (int s = 0; s <= g; ++s) acc[1][1][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(dav[s], dbv[g - s], acc[1][1][g], 0, 0, 0);
    }
    v4i bv[2][4], av[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[0][s] = *(const v4i*)(sB + s * 2048 + (lane & 31) * 32 + (lane >> 5) * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) av[0][s] = *(const v4i*)(sA + s * 8192 + (w * 64 + (lane & 31)) * 32 + (lane >> 5) * 16);
    if (V != 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[1][s] = *(const v4i*)(sB + s * 2048 + (32 + (lane & 31)) * 32 + (lane >> 5) * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) av[1][s] = *(const v4i*)(sA + s * 8192 + (w * 64 + 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
    }
    int nm = 0, nd = 0;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (V == 1 && f == 0 && h == 1) {
#pragma unroll
          for (int s = 0; s < 4; ++s) bv[1][s] = *(const v4i*)(sB + s * 2048 + (32 + (lane & 31)) * 32 + (lane >> 5) * 16);
#pragma unroll
          for (int s = 0; s < 4; ++s) av[1][s] = *(const v4i*)(sA + s * 8192 + (w * 64 + 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (V == 3 && f == 1 && h == 1 && g >= 2) continue;
#pragma unroll
          for (int s = 0; s <= g; ++s) {
            acc[f][h][g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[f][s], bv[h][g - s], acc[f][h][g], 0, 0, 0);
            if (V != 2 && (++nm & 3) == 0 && nd < 10) {
              glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + (w * 10 + nd) * 1024));
              ++nd;
            }
          }
        }
      }
    if (V == 3) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { dav[s] = av[1][s]; dbv[s] = bv[1][s]; }
      while (nd < 10) { glds16(next_ptr(), __builtin_amdgcn_readfirstlane(lds0 + ((it + 2) % 3) * 40960 + (w * 10 + nd) * 1024)); ++nd; }
    }
    if (V != 2) wait_vm<10>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  int sum = 0;
  for (int f = 0; f < 2; ++f) for (int h = 0; h < 2; ++h) for (int g = 0; g < 4; ++g) sum += acc[f][h][g][0] + acc[f][h][g][9];
  if (sum == 0x12345678) sink[tid] = sum;
  if (lane == 0) out[blockIdx.x * NW + w] = t1 - t0;
}
template <int V>
void run512(const unsigned char* buf, unsigned long long* out, int* sink, const char* what) {
  hipFuncSetAttribute((const void*)k512<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k512<V><<<256, 256, 140 * 1024>>>(buf, out, sink);
  hipEventRecord(e0);
  k512<V><<<256, 256, 140 * 1024>>>(buf, out, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * NW);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0;
  for (int blk = 0; blk < 256; ++blk) for (int w = 0; w < 4; ++w) a += (double)h[blk * NW + w];
  printf("512-register form %d  %-66s kernel %7.3f ms = %6.0f ns/iter | cycles/iter %6.0f\n", V, what, ms, ms * 1e6 / ITER, a / (256 * 4.0 * ITER));
  fflush(stdout);
}

template <int MODE>
void run(const unsigned char* buf, unsigned long long* out, int* sink, const char* what) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512, 140 * 1024>>>(buf, out, sink);
  hipEventRecord(e0);
  k<MODE><<<256, 512, 140 * 1024>>>(buf, out, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * NW);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int blk = 0; blk < 256; ++blk) for (int w = 0; w < NW; ++w) (w < 4 ? a : b) += (double)h[blk * NW + w];
  a /= 256 * 4.0 * ITER; b /= 256 * 4.0 * ITER;
  if (hipGetLastError() != hipSuccess) printf("mode %d: HIP error\n", MODE);
  printf("mode %d  %-78s kernel %7.3f ms = %6.0f ns/iter | cycles/iter waves 0-3: %6.0f  waves 4-7: %6.0f\n", MODE, what, ms,
         ms * 1e6 / ITER, a, b);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
#define RUN(M, what) if (only < 0 || only == M) run<M>(buf, out, sink, what)
  unsigned char* buf; unsigned long long* out; int* sink;
  hipMalloc(&buf, BUF + (1 << 20)); hipMemset(buf, 1, BUF + (1 << 20));
  hipMalloc(&out, 256 * NW * 8); hipMalloc(&sink, 4096);
  RUN(0, "all waves: 4 plain global_load_dwordx4 / iter (32 KiB per CU-iter)");
  RUN(1, "all waves: 4 global_load_lds_dwordx4 / iter, m0 saved + restored");
  RUN(7, "all waves: 4 global_load_lds_dwordx4 / iter, m0 written only");
  RUN(2, "all waves: 12 ds_read_b128 + 20 i8 MFMA / iter + barrier (floor 1280)");
  RUN(3, "waves 0-3: 10 LDS-DMA / iter | waves 4-7: compute (floor 640)");
  RUN(4, "waves 0-3: 10 plain loads / iter | waves 4-7: compute (floor 640)");
  RUN(5, "all waves: 1 DMA + 4 plain (A in registers) + 8 ds_read + 20 MFMA + barrier");
  RUN(6, "all waves: 5 DMA, then 12 ds_read + 20 MFMA + barrier (the shipped step)");
  RUN(8, "all waves: 12 ds_read + 20 MFMA with the 5 DMA between the fragments + barrier");
  if (only == 20) {
    run_ov<1, true>(out, sink, "waves 0-3: 20 i8 MFMA / iter alone");
    run_ov<2, true>(out, sink, "waves 4-7: 320 v_fma_f64 / iter alone");
    run_ov<3, true>(out, sink, "both: does f64 VALU overlap the SIMD partner's MFMAs?");
    run_ov<2, false>(out, sink, "waves 4-7: 320 v_fma_f32 / iter alone");
    run_ov<3, false>(out, sink, "both, f32");
  }
  if (only == 30) {
    run512<2>(buf, out, sink, "4 waves, 64 x 64 wave tiles: 16 ds_read + 40 MFMA + barrier (no DMA)");
    run512<0>(buf, out, sink, "... + 10 DMA per wave, one after every fourth MFMA");
    run512<1>(buf, out, sink, "... second fragments' reads after the first ten MFMAs");
    run512<3>(buf, out, sink, "... last eight MFMAs deferred behind the barrier");
  }
  RUN(9, "as 8, one barrier per TWO iterations");
  RUN(10, "12 ds_read + 20 MFMA, one DMA after every fourth MFMA + barrier");
  RUN(11, "20 MFMA only (floor 1280)");
  RUN(12, "as 8 without the barrier");
  RUN(13, "as 10, both fragments' A operands requested up front");
  RUN(16, "A operand in REGISTERS (plain loads, two iterations ahead), B via LDS; traffic between the B fragments + barrier");
  RUN(17, "as 16, the traffic before the MFMAs");
  RUN(14, "as 8, the last 4 MFMAs of an iteration deferred behind the barrier");
  RUN(15, "as 8, the last 7 MFMAs of an iteration deferred behind the barrier");
  return 0;
}
