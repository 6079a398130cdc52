// Development aid: single-wave issue cost / latency of the instructions of the factor leaf's pivot chain (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned long long g_t[32];
__device__ double g_sink[64];
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ void k(double x0) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = x0 + i + threadIdx.x;
  double y = x0 * 0.5, z = x0 + 2.0;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
    if (MODE == 0) {  // 16 independent plain fmac
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(z));
    } else if (MODE == 1) {  // 16 independent dpp fmac
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y), "v"(z));
    } else if (MODE == 2) {  // dependent plain fmac chain
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[0]) : "v"(y), "v"(z));
    } else if (MODE == 3) {  // dependent mov_dpp chain (with the 2 wait states)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
    } else if (MODE == 4) {  // dependent rsq chain
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f64_e32 %0, %0" : "+v"(a[0]));
    } else if (MODE == 5) {  // dependent mul chain
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[0]) : "v"(y));
    } else if (MODE == 6) {  // 16 independent mov_dpp (no nop)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(y));
    } else if (MODE == 7) {  // independent readlane pairs
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3" :: "v"(__double2loint(a[i])), "v"(__double2hiint(a[i])) : "s20", "s21");
    } else if (MODE == 8) {  // independent 32-bit fma
#pragma unroll
      for (int i = 0; i < 16; ++i) { float fa = (float)a[i]; asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(fa) : "v"((float)y), "v"((float)z)); a[i] = fa; }
    } else if (MODE == 9) {  // dependent dpp fmac chain through the DPP operand: fmac -> (nop) -> fmac reading it via dpp
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(z));
    } else if (MODE == 10) {  // round 6: the other instructions of the int8 sweep's K* generation, 16 independent each
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f64_e32 %0, %1" : "=v"(a[i]) : "v"(y));
    } else if (MODE == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rndne_f64_e32 %0, %1" : "=v"(a[i]) : "v"(y));
    } else if (MODE == 12) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { int r; asm volatile("v_cvt_i32_f64_e32 %0, %1" : "=v"(r) : "v"(a[i])); a[i] = __hiloint2double(r, r); }
    } else if (MODE == 13) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(a[i]) : "v"(y), "v"(3));
    } else if (MODE == 14) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f64 %0, %1, %2" : "=v"(a[i]) : "v"(y), "v"(z));
    } else if (MODE == 15) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(y), "v"(z));
    } else if (MODE == 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_i32_e32 %0, %1" : "=v"(a[i]) : "v"(it));
    } else if (MODE == 17) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { int r; asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(it), "v"(i), "v"(0x05010400)); a[i] = __hiloint2double(r, r); }
    } else if (MODE == 18) {  // 64-bit integer shift-add (the digits of the fold)
#pragma unroll
      for (int i = 0; i < 16; ++i) { long long r; asm volatile("v_lshl_add_u64 %0, %1, 8, %2" : "=v"(r) : "v"((long long)it), "v"((long long)i)); a[i] = __longlong_as_double(r); }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  g_sink[threadIdx.x] = s;
  if (threadIdx.x == 0) g_t[MODE] = t1 - t0;
}
int main() {
  k<0><<<1, 64>>>(1.5); k<1><<<1, 64>>>(1.5); k<2><<<1, 64>>>(1.5); k<3><<<1, 64>>>(1.5); k<4><<<1, 64>>>(1.5);
  k<5><<<1, 64>>>(1.5); k<6><<<1, 64>>>(1.5); k<7><<<1, 64>>>(1.5); k<8><<<1, 64>>>(1.5); k<9><<<1, 64>>>(1.5);
  k<10><<<1, 64>>>(1.5); k<11><<<1, 64>>>(1.5); k<12><<<1, 64>>>(1.5); k<13><<<1, 64>>>(1.5); k<14><<<1, 64>>>(1.5);
  k<15><<<1, 64>>>(1.5); k<16><<<1, 64>>>(1.5); k<17><<<1, 64>>>(1.5); k<18><<<1, 64>>>(1.5);
  hipDeviceSynchronize();
  unsigned long long t[32];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
  const char* names[] = {"independent v_fmac_f64", "independent v_fmac_f64_dpp", "dependent v_fmac_f64", "dependent nop+v_mov_b64_dpp",
                         "dependent v_rsq_f64", "dependent v_mul_f64", "independent v_mov_b64_dpp", "independent readlane pair",
                         "independent v_fmac_f32", "dependent nop+v_fmac_f64_dpp (via dpp src)", "independent v_rsq_f64", "independent v_rndne_f64",
                         "independent v_cvt_i32_f64", "independent v_ldexp_f64", "independent v_max_f64", "independent v_add_f64",
                         "independent v_cvt_f64_i32", "independent v_perm_b32", "independent v_lshl_add_u64"};
  for (int m = 0; m < 19; ++m) printf("%-44s %.1f ticks per instruction (256 instructions, s_memtime)\n", names[m], t[m] / 256.0);
  return 0;
}
