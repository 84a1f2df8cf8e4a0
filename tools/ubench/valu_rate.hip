// VALU issue-rate probe for gfx950: wave-level instructions per second for a few integer opcodes the extractor's kernels are
// made of, at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void k(unsigned* out, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  const unsigned b = blockIdx.x | 0x01010101u;
  for (int i = 0; i < iters; i++) {
#define STEP(r)                                                                                      \
    if (OP == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(b));                         \
    else if (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r) : "v"(b));                  \
    else if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));                       \
    else if (OP == 3) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(r) : "v"(b));               \
    else if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r) : "v"(b));                  \
    else if (OP == 5) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(r) : "v"(b));                  \
    else if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r) : "v"(b));               \
    else if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(b));                    \
    else if (OP == 8) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r) : "v"(b));         \
    else if (OP == 9) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(r) : "v"(b));                   \
    else if (OP == 10) asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(r) : "v"(b));                  \
    else if (OP == 11) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(r));                        \
    else if (OP == 12) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(r) : "v"(b));                  \
    else if (OP == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r) : "v"(b));               \
    else if (OP == 14) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(b));                   \
    else if (OP == 15) asm volatile("v_max_i16_e32 %0, %0, %1" : "+v"(r) : "v"(b));                 \
    else if (OP == 16) asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_1" : "+v"(r) : "v"(b)); \
    else if (OP == 17) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(r) : "v"(b));                 \
    else if (OP == 18) asm volatile("v_min_u16_e32 %0, %0, %1" : "+v"(r) : "v"(b));                 \
    else if (OP == 19) asm volatile("v_sub_u16_e32 %0, %0, %1" : "+v"(r) : "v"(b));                 \
    else if (OP == 20) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(r) : "v"(b));                  \
    else if (OP == 21) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %1" : "+v"(r) : "v"(b));         \
    else if (OP == 22) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(r));                       \
    else if (OP == 23) asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_1" : "+v"(r) : "v"(b)); \
    else if (OP == 24) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r) : "v"(b) : );    \
    else if (OP == 25) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r)); \
    else if (OP == 26) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(r) : "v"(b));                 \
    else if (OP == 27) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(r) : "v"(b));
    STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
  }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;
}

template <int OP>
int run(const char* name, unsigned* d) {
  const int iters = 4096;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;  // 256 CUs x (4 SIMDs x wps waves) / 4 waves per block
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
    printf("%-16s %d waves/SIMD: %7.1f G wave-instr/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, wps, winstr / ms / 1e6,
           1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
  }
  return 0;
}

int main() {
  unsigned* d;
  CK(hipMalloc(&d, 64));
  run<0>("v_pk_max_i16", d); run<1>("v_perm_b32", d); run<2>("v_add_u32", d); run<3>("v_dot4_u32_u8", d); run<4>("v_bcnt_u32_b32", d);
  run<5>("v_min3_u32", d); run<6>("v_mad_u32_u24", d); run<7>("v_mul_lo_u32", d); run<8>("v_alignbyte_b32", d);
  run<9>("v_xor_b32_e32", d); run<10>("v_max_i32_e32", d); run<11>("v_lshlrev_b32_e32", d); run<12>("v_xor_b32_e64", d);
  run<13>("v_lshl_or_b32", d); run<14>("v_pk_add_u16", d); run<15>("v_max_i16_e32", d);
  run<16>("v_min_u16_sdwa(b,w1)", d); run<17>("v_and_b32_e32", d); run<18>("v_min_u16_e32", d); run<19>("v_sub_u16_e32", d);
  run<20>("v_pk_min_i16", d); run<21>("v_pk_minimum3_f16", d); run<22>("v_lshrrev_b32", d); run<23>("v_min_u16_sdwa(pad)", d);
  run<24>("v_cndmask_b32", d); run<25>("v_add_u32_dpp", d); run<26>("v_sub_u32", d); run<27>("v_or_b32", d);
  return 0;
}
