// Development aid: phase timing of the 128-leaf (csrc/tgp_kernels_leaf.hip) with s_memrealtime stamps (100 MHz).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I trieste_amd/csrc -I include tools/ubench_leaf.hip -o tools/ubench_leaf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ unsigned long long g_ticks[96];
#define TGP_LEAF_TICK(i) do { if (threadIdx.x == 0) g_ticks[i] = __builtin_amdgcn_s_memtime(); } while (0)
#include "tgp_kernels_leaf.hip"

int main() {
  const int n = 128, ld = 128;
  std::vector<double> A(n * n), L(n * n), W(n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.05 * (i - j) * (i - j)) + (i == j ? 0.1 : 0.0);
  double *dA, *dL, *dW; int* dinfo;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dW, n * n * 8); hipMalloc(&dinfo, 4);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    for (int it = 0; it < 10; ++it) tgp::launch_leaf128(0, dA, dL, dW, ld, 0, dinfo);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("10 dependent launches: %.1f us each\n", 100.0 * ms);
  }
  unsigned long long t[96];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), sizeof(t));
  hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(W.data(), dW, n * n * 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double s = 0; for (int k = 0; k < n; ++k) s += W[i * n + k] * L[k * n + j];
    err = std::fmax(err, std::fabs(s - (i == j)));
  }
  printf("|W L - I| = %.2e\n", err);
  auto us = [&](int a, int b) { return (double)(t[b] - t[a]); };
  printf("load..start: (tick 0); total tick0->33 %.0f\n", us(0, 33));
  for (int kb = 0; kb < 8; ++kb) {
    const int prev = kb == 0 ? 0 : 4 * kb;
    printf("kb %d: panel %.0f  barrier %.0f  phaseA %.0f  phaseB %.0f\n", kb, us(prev, 1 + 4 * kb), us(1 + 4 * kb, 2 + 4 * kb),
           us(2 + 4 * kb, 3 + 4 * kb), us(3 + 4 * kb, 4 + 4 * kb));
  }
  printf("store %.0f\n", us(32, 33));


  return 0;
}
