// VALU issue-rate probe, float side: scalar-per-lane fp32 ops against the PACKED fp32 forms (two floats per instruction on a register
// pair) of gfx950, and the mbcnt / cmp_sdwa / readlane forms k_fast_wave leans on.  hipcc --offload-arch=gfx950 -O3 valu_rate_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void k(unsigned* out, int iters) {
  typedef unsigned long long u64;
  u64 q0 = threadIdx.x | 0x3f80000000000000ull, q1 = q0 + 3, q2 = q0 + 5, q3 = q0 + 7, q4 = q0 + 11, q5 = q0 + 13, q6 = q0 + 17, q7 = q0 + 19;
  unsigned a0 = threadIdx.x | 0x3f800000u, a1 = a0 + 3, a2 = a0 + 5, a3 = a0 + 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  const u64 bq = 0x3f8000013f800001ull;
  const unsigned b = 0x3f800001u;
  for (int i = 0; i < iters; i++) {
#define STEP(r, q)                                                                                       \
    if (OP == 0) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(r) : "v"(b));                             \
    else if (OP == 1) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(r) : "v"(b));                        \
    else if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q) : "v"(bq));                        \
    else if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q) : "v"(bq));                        \
    else if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(b));                        \
    else if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q) : "v"(bq));                    \
    else if (OP == 6) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(r) : "v"(b));                    \
    else if (OP == 7) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r) : "v"(b));
    STEP(a0, q0) STEP(a1, q1) STEP(a2, q2) STEP(a3, q3) STEP(a4, q4) STEP(a5, q5) STEP(a6, q6) STEP(a7, q7)
    STEP(a0, q0) STEP(a1, q1) STEP(a2, q2) STEP(a3, q3) STEP(a4, q4) STEP(a5, q5) STEP(a6, q6) STEP(a7, q7)
  }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u || (q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) == 0x12345ull) out[0] = a0;
}

template <int OP>
int run(const char* name, unsigned* d) {
  const int iters = 4096;
  for (int wps : {2, 8}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * 4 * iters * 16;
    printf("%-18s %d waves/SIMD: %7.1f G wave-instr/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, wps, winstr / ms / 1e6,
           1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
  }
  return 0;
}

int main() {
  unsigned* d;
  CK(hipMalloc(&d, 64));
  run<0>("v_mul_f32", d); run<1>("v_add_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d); run<4>("v_fma_f32", d);
  run<5>("v_pk_fma_f32", d); run<6>("v_mad_i32_i24", d); run<7>("v_mbcnt_lo_u32_b32", d);
  return 0;
}
