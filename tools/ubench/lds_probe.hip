#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint2 __attribute__((aligned(2))) uint2u;
typedef uint32_t __attribute__((aligned(1))) u32u;
__global__ void probe(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + lane;
  uint32_t v = 0xAAAAAAAAu;
  asm volatile("ds_read_u8_d16 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_read_u8_d16_hi %0, %1 offset:5\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(addr) : "memory");
  out[lane] = v;
  uint32_t w = 0xAAAAAAAAu;
  asm volatile("ds_read_u8_d16_hi %0, %1 offset:5\n s_waitcnt lgkmcnt(0)" : "+v"(w) : "v"(addr) : "memory");
  out[64 + lane] = w;
  // unaligned b64 at 2-byte alignment and b32 at odd addresses
  uint2 q;
  uint32_t a2 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + 2 * lane + 6;
  asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a2) : "memory");
  out[128 + 2 * lane] = q.x; out[128 + 2 * lane + 1] = q.y;
  uint32_t d;
  uint32_t a1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + 3 * lane + 1;
  asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a1) : "memory");
  out[256 + lane] = d;
}
__global__ void cc(const uint8_t* p, uint2* o, uint32_t* o2) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = p[i];
  __syncthreads();
  o[threadIdx.x] = *reinterpret_cast<const uint2u*>(lds + 2 * threadIdx.x + 6);
  o2[threadIdx.x] = *reinterpret_cast<const u32u*>(lds + 3 * threadIdx.x + 1);
}
int main() {
  uint32_t* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
  auto L = [](int i) { return (uint32_t)(uint8_t)(i * 7 + 3); };
  int ok1 = 0, ok1b = 0, ok2 = 0, ok3 = 0;
  for (int l = 0; l < 64; l++) {
    if (h[l] == (L(l) | (L(l + 5) << 16))) ok1++;
    if (h[64 + l] == (0xAAAAu | (L(l + 5) << 16))) ok1b++;
    uint32_t e0 = 0, e1 = 0; for (int b = 0; b < 4; b++) { e0 |= L(2 * l + 6 + b) << (8 * b); e1 |= L(2 * l + 10 + b) << (8 * b); }
    if (h[128 + 2 * l] == e0 && h[129 + 2 * l] == e1) ok2++;
    uint32_t e = 0; for (int b = 0; b < 4; b++) e |= L(3 * l + 1 + b) << (8 * b);
    if (h[256 + l] == e) ok3++;
  }
  printf("d16 lo+hi packs: %d/64 (lane0 %08x)  d16_hi preserves low: %d/64 (lane0 %08x)  b64@2: %d/64  b32@odd: %d/64\n", ok1, h[0], ok1b, h[64], ok2, ok3);
  return 0;
}
