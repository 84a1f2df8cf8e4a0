// Does LDS-DMA (global_load_lds_dword) accept global addresses that are not dword aligned on gfx950?
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ldsdma tools/ubench/lds_dma_unaligned.hip ; run: /tmp/ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string.h>
__global__ void k(const uint8_t* src, uint32_t* out, int shift) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + 4 * lane),
                                   (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[lane] = lds[lane];
}
int main() {
  std::vector<uint8_t> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = (uint8_t)(i * 7 + 3);
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, 1024); hipMalloc(&o, 256);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  int bad = 0;
  for (int shift = 0; shift < 8; shift++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
    uint32_t r[64];
    if (hipMemcpy(r, o, 256, hipMemcpyDeviceToHost) != hipSuccess) { printf("shift %d: launch failed\n", shift); return 1; }
    int ok = 1;
    for (int l = 0; l < 64; l++) { uint32_t e; memcpy(&e, &h[shift + 4 * l], 4); if (e != r[l]) ok = 0; }
    printf("shift %d: %s\n", shift, ok ? "exact" : "MISMATCH");
    bad += !ok;
  }
  return bad;
}
