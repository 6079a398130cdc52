// Controlled ladder for the u16 sweep structure on MI355X: which ingredient costs the MFMA pipe what?
// 1024-thread workgroup (16 waves, 4 per SIMD), wave tile 64x32 (acc[4][2]), BK = 16 per step,
// double-buffered LDS stages of [16][272] + [16][144] doubles, one workgroup per CU.
//   mode bit0: __syncthreads() per step          bit1: LDS writes of the staged tile (48 KB / step)
//   mode bit2: global loads of the tile (Wt-like 32 KB shared by all workgroups + 16 KB private slab)
//   mode bit3: the loaded registers are what gets written (vmcnt wait before the LDS writes)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_ladder.hip -o tools/ubench_ladder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int LDA = 272, LDB = 144, STAGE = 16 * LDA + 16 * LDB;

template <int MODE>
__global__ __launch_bounds__(1024, 4) void ladder(double* out, const double* __restrict__ W, const double* __restrict__ slab,
                                                  int steps, int64_t wstride) {
  __shared__ __attribute__((aligned(16))) double smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 2, wn = w & 3;
  for (int i = tid; i < 2 * STAGE; i += 1024) smem[i] = 1e-3 * (i % 97);
  __syncthreads();
  v4d acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0, 0, 0, 0};
  const int wrow0 = tid >> 7, wcol = (tid & 127) * 2, kcol = tid & 127, krg = tid >> 7;
  const double* myslab = slab + (size_t)blockIdx.x * 16 * 128 * 64;
  v2d wreg0 = {1e-3, 2e-3}, wreg1 = {3e-3, 4e-3};
  double kv0 = 1e-3, kv1 = 2e-3;
  int buf = 0;
  for (int t = 0; t < steps; ++t) {
    if (MODE & 4) {
      const double* src = W + (size_t)(t % 240) * 16 * wstride + (size_t)wrow0 * wstride + wcol;
      const v2d a0 = *(const v2d*)src, a1 = *(const v2d*)(src + 8 * wstride);
      const double* kp = myslab + (size_t)(t % 64) * 16 * 128 + (2 * krg) * 128 + kcol;
      const double b0 = kp[0], b1 = kp[128];
      if (MODE & 8) { wreg0 = a0; wreg1 = a1; kv0 = b0; kv1 = b1; }
      else { asm volatile("" :: "v"(a0), "v"(a1), "v"(b0), "v"(b1)); }
    }
    {
      const double* sA = smem + buf * STAGE;
      const double* ab = sA + (lane >> 4) * LDA + wm * 64 + (lane & 15);
      const double* bb = sA + 16 * LDA + (lane >> 4) * LDB + wn * 32 + (lane & 15);
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        double av[4], bv[2];
#pragma unroll
        for (int f = 0; f < 4; ++f) av[f] = ab[k4 * 4 * LDA + f * 16];
#pragma unroll
        for (int f = 0; f < 2; ++f) bv[f] = bb[k4 * 4 * LDB + f * 16];
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
          for (int fn = 0; fn < 2; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[fm], bv[fn], acc[fm][fn], 0, 0, 0);
      }
    }
    if (MODE & 2) {
      double* sA = smem + (buf ^ 1) * STAGE;
      *(v2d*)(sA + wrow0 * LDA + wcol) = wreg0;
      *(v2d*)(sA + (wrow0 + 8) * LDA + wcol) = wreg1;
      sA[16 * LDA + (2 * krg) * LDB + kcol] = kv0;
      sA[16 * LDA + (2 * krg + 1) * LDB + kcol] = kv1;
    }
    if (MODE & 1) __syncthreads();
    else asm volatile("" ::: "memory");
    buf ^= 1;
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 1024 + tid] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int MODE> void run(const char* name, double* out, const double* W, const double* slab, int cu, int steps, int64_t ws) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(ladder<MODE>, dim3(cu), dim3(1024), 0, 0, out, W, slab, steps, ws); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(ladder<MODE>, dim3(cu), dim3(1024), 0, 0, out, W, slab, steps, ws); CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)cu * 16 * steps * 32 * 2048.0;
  printf("%-58s %8.2f ms  %6.2f TFLOP/s  (%.0f cycles/step @2.4GHz)\n", name, ms, flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / steps);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cu = p.multiProcessorCount, steps = 4000; const int64_t ws = 4096;
  double *out, *W, *slab;
  CK(hipMalloc(&out, sizeof(double) * cu * 1024));
  CK(hipMalloc(&W, sizeof(double) * ws * 4096)); CK(hipMemset(W, 0, sizeof(double) * ws * 4096));
  CK(hipMalloc(&slab, sizeof(double) * (size_t)cu * 16 * 128 * 64)); CK(hipMemset(slab, 0, sizeof(double) * (size_t)cu * 16 * 128 * 64));
  run<0>("L0 LDS-fed MFMA only", out, W, slab, cu, steps, ws);
  run<1>("L1 + barrier per step", out, W, slab, cu, steps, ws);
  run<2>("   LDS writes, no barrier", out, W, slab, cu, steps, ws);
  run<3>("L2 barrier + LDS writes", out, W, slab, cu, steps, ws);
  run<4>("   global loads only (unused), no barrier", out, W, slab, cu, steps, ws);
  run<5>("   barrier + global loads (unused)", out, W, slab, cu, steps, ws);
  run<7>("L3 barrier + LDS writes + global loads (unused)", out, W, slab, cu, steps, ws);
  run<15>("L4 barrier + loads -> vmcnt -> LDS writes (real staging)", out, W, slab, cu, steps, ws);
  run<14>("   loads -> LDS writes, no barrier", out, W, slab, cu, steps, ws);
  return 0;
}
