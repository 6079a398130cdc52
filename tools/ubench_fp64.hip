// fp64 micro-benchmarks for MI355X (gfx950): what the sweep kernel's roofline really is.
//   1. v_mfma_f64_16x16x4_f64 peak            2. v_fma_f64 (VALU) peak
//   3. both from ONE wave (interleaved)        4. both from DIFFERENT waves sharing a SIMD
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NM, int NF>  // per iteration: NM mfma (16 independent accs) + NF fma
__global__ __launch_bounds__(256) void k_same(double* out, int iters, double a, double b) {
  v4d acc[16];
  double f[16];
  for (int i = 0; i < 16; ++i) { acc[i] = (v4d){0, 0, 0, 0}; f[i] = threadIdx.x * 1e-9 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < NM) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NF / 16; ++q) f[(r + q) & 15] = __builtin_fma(f[(r + q) & 15], a, b);
    }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves with (wave & 1) == 0 run MFMA, others run FMA  (block of 512 = 8 waves, 2 per SIMD)
__global__ __launch_bounds__(512) void k_split(double* out, int iters, double a, double b, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 ? true : mode == 1 ? false : ((wave >> 2) == 0);
  double s = 0;
  if (do_mfma) {
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v4d){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r], 0, 0, 0);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) f[q] = __builtin_fma(f[q], a, b);
    for (int i = 0; i < 16; ++i) s += f[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// 768-thread block = 12 waves = 3 per SIMD.  mode 0: waves 0-7 MFMA, 8-11 idle; mode 1: 0-7 idle,
// 8-11 FMA (fiters); mode 2: both; mode 3: all 12 waves MFMA.
__global__ __launch_bounds__(768) void k_12(double* out, int iters, int fiters, double a, double b, int mode) {
  const int wave = threadIdx.x >> 6;
  double s = 0;
  const bool mf = (mode == 3) || (wave < 8 && (mode == 0 || mode == 2));
  const bool ff = (mode != 3) && (wave >= 8 && (mode == 1 || mode == 2));
  if (mf) {
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v4d){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r], 0, 0, 0);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (ff) {
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < fiters; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) f[q] = __builtin_fma(f[q], a, b);
    for (int i = 0; i < 16; ++i) s += f[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS-fed MFMA loop like the sweep consumer: per k4 step 4 A + 4 B ds_read_b64 and 16 MFMA, no barrier.
__global__ __launch_bounds__(512) void k_ldsfed(double* out, int iters) {
  __shared__ double As[16 * 272];
  __shared__ double Bs[16 * 144];
  for (int i = threadIdx.x; i < 16 * 272; i += 512) As[i] = 1e-3 * i;
  for (int i = threadIdx.x; i < 16 * 144; i += 512) Bs[i] = 1e-3 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1;
  v4d acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0, 0, 0, 0};
  const double* ab = As + (lane >> 4) * 272 + wm * 64 + (lane & 15);
  const double* bb = Bs + (lane >> 4) * 144 + wn * 64 + (lane & 15);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      double av[4], bv[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) { av[f] = ab[k4 * 4 * 272 + f * 16]; bv[f] = bb[k4 * 4 * 144 + f * 16]; }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[fm], bv[fn], acc[fm][fn], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// transcendental-ish costs: rsq / rcp / ldexp / rndne throughput
__global__ __launch_bounds__(256) void k_trans(double* out, int iters, double a, int which) {
  double f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.0 + threadIdx.x * 1e-6 + i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (which == 0) f[q] = __builtin_amdgcn_rsq(f[q]) + a;
      else if (which == 1) f[q] = __builtin_amdgcn_rcp(f[q]) + a;
      else if (which == 2) f[q] = __builtin_rint(f[q] * a) + a;
      else f[q] = f[q] * a;  // v_mul_f64
    }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class F> double time_ms(F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d MHz  LDS/block %zu  L2 %d MiB  regs/block %d\n", p.name, p.multiProcessorCount,
         p.clockRate / 1000, p.sharedMemPerBlock, p.l2CacheSize >> 20, p.regsPerBlock);
  const int CU = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, sizeof(double) * CU * 8 * 512));
  const int iters = 20000;
  for (int occ : {1, 2}) {
    const int grid = CU * occ;  // 256-thread blocks: 4 waves -> occ waves per SIMD
    const double waves = (double)grid * 4;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL((k_same<16, 0>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("[occ %d] mfma only        : %8.2f TFLOP/s (mfma)\n", occ, waves * iters * 16 * 2048.0 / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_same<0, 256>), dim3(grid), dim3(256), 0, 0, out, iters / 4, 1.0000001, 1e-9); });
    printf("[occ %d] fma only         : %8.2f TFLOP/s (valu)\n", occ, waves * (iters / 4) * 256 * 128.0 / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_same<16, 64>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("[occ %d] same wave 16m+64f: %8.2f TFLOP/s mfma + %8.2f valu (ms %.2f)\n", occ, waves * iters * 16 * 2048.0 / ms * 1e-9,
           waves * iters * 64 * 128.0 / ms * 1e-9, ms);
    ms = time_ms([&] { hipLaunchKernelGGL((k_same<16, 128>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("[occ %d] same wave 16m+128f: %7.2f TFLOP/s mfma + %8.2f valu (ms %.2f)\n", occ, waves * iters * 16 * 2048.0 / ms * 1e-9,
           waves * iters * 128 * 128.0 / ms * 1e-9, ms);
    ms = time_ms([&] { hipLaunchKernelGGL((k_same<16, 256>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("[occ %d] same wave 16m+256f: %7.2f TFLOP/s mfma + %8.2f valu (ms %.2f)\n", occ, waves * iters * 16 * 2048.0 / ms * 1e-9,
           waves * iters * 256 * 128.0 / ms * 1e-9, ms);
  }
  {
    const int grid = CU;  // 512-thread blocks: 8 waves, 2 per SIMD
    double ms0 = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(grid), dim3(512), 0, 0, out, iters, 1.0000001, 1e-9, 0); });
    double ms1 = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(grid), dim3(512), 0, 0, out, iters / 4, 1.0000001, 1e-9, 1); });
    printf("[split] all-mfma 8 waves  : %8.2f TFLOP/s (ms %.2f)\n", (double)grid * 8 * iters * 16 * 2048.0 / ms0 * 1e-9, ms0);
    printf("[split] all-fma  8 waves  : %8.2f TFLOP/s (ms %.2f)\n", (double)grid * 8 * (iters / 4) * 256 * 128.0 / ms1 * 1e-9, ms1);
    // half the waves mfma (iters x 16 mfma = iters*1024 cycles each), half fma (iters x 256 fma)
    double ms2 = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(grid), dim3(512), 0, 0, out, iters, 1.0000001, 1e-9, 2); });
    printf("[split] 4 mfma + 4 fma waves: mfma %8.2f TFLOP/s + valu %8.2f TFLOP/s (ms %.2f; mfma-alone would take %.2f, fma-alone %.2f)\n",
           (double)grid * 4 * iters * 16 * 2048.0 / ms2 * 1e-9, (double)grid * 4 * iters * 256 * 128.0 / ms2 * 1e-9, ms2, ms0 / 2,
           ms1 * 4 / 2);
  }

  {
    const int grid = CU;
    const int it = 20000, fit = 1000;   // FMA waves alone: fit*256 fma
    double m0 = time_ms([&] { hipLaunchKernelGGL(k_12, dim3(grid), dim3(768), 0, 0, out, it, fit, 1.0000001, 1e-9, 0); });
    double m1 = time_ms([&] { hipLaunchKernelGGL(k_12, dim3(grid), dim3(768), 0, 0, out, it, fit, 1.0000001, 1e-9, 1); });
    double m3 = time_ms([&] { hipLaunchKernelGGL(k_12, dim3(grid), dim3(768), 0, 0, out, it, fit, 1.0000001, 1e-9, 3); });
    printf("[12w] 8 mfma waves alone %.2f ms (%.2f TF); 4 fma waves alone (fit=%d) %.2f ms; 12 mfma waves %.2f ms (%.2f TF)\n", m0,
           (double)grid * 8 * it * 16 * 2048.0 / m0 * 1e-9, fit, m1, m3, (double)grid * 12 * it * 16 * 2048.0 / m3 * 1e-9);
    for (int f : {500, 1000, 1500, 2000, 3000}) {
      double ma = time_ms([&] { hipLaunchKernelGGL(k_12, dim3(grid), dim3(768), 0, 0, out, it, f, 1.0000001, 1e-9, 1); });
      double m2 = time_ms([&] { hipLaunchKernelGGL(k_12, dim3(grid), dim3(768), 0, 0, out, it, f, 1.0000001, 1e-9, 2); });
      printf("[12w] 8 mfma + 4 fma(fit=%d): together %.2f ms ; mfma alone %.2f, fma alone %.2f -> sum %.2f\n", f, m2, m0, ma, m0 + ma);
    }
    double ml = time_ms([&] { hipLaunchKernelGGL(k_ldsfed, dim3(grid), dim3(512), 0, 0, out, 5000); });
    printf("[ldsfed] 8 waves LDS-fed 64x64 tiles: %.2f TFLOP/s (ms %.2f)\n", (double)grid * 8 * 5000 * 64 * 2048.0 / ml * 1e-9, ml);
  }
  for (int which = 0; which < 0; ++which) {
    const int grid = CU * 2;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_trans, dim3(grid), dim3(256), 0, 0, out, 20000, 1.0000001, which); });
    const char* nm[] = {"rsq+add", "rcp+add", "rndne(mul)+add", "mul"};
    printf("[valu] %-15s: %.2f cycles per wave-instruction-pair per SIMD (@%d MHz)\n", nm[which],
           ms * 1e-3 * p.clockRate * 1e3 / (20000.0 * 8 * 2), p.clockRate / 1000);
  }
  return 0;
}
