// tools/probe/mfma_f8_probe.hip — what v_mfma_scale_f32_32x32x64_f8f6f4 computes for the byte encodings the match wants (round 5):
// A (train) in bf8 with byte 0x04 = 2^-14 per set bit, B (query) in bf8 with byte 0x58 = 2^7: every common bit adds 2^-7;
// a second product with A in fp8 e4m3 (small signed integers) against bf8 constants.  Assumed layouts, checked here against a host sum:
// operand lane l = 32 h + r supplies 32 k-values of row r as its 32 bytes; D register i of lane l = row (i & 3) + 8 (i >> 2) + 4 (l >> 5),
// column l & 31.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f8_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_probe(const v8i* a, const v8i* b, float* d, int cbsz_sel) {
  const int l = threadIdx.x;
  v16f acc = {0};
  if (cbsz_sel == 2) {   // A, B fp4 (e2m1) nibbles, B scaled by 2^7 through its block scale; C = the lane's register index (a non-zero C operand)
    for (int i = 0; i < 16; i++) acc[i] = (float)(100 * i);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x86868686);
  } else if (cbsz_sel == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // A bf8, B bf8
  else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                // A fp8 (e4m3), B bf8
  for (int i = 0; i < 16; i++) d[l * 16 + i] = acc[i];
}
template <int F>
__global__ void k_time_f(const v8i* a, const v8i* b, float* d, int iters) {   // F = 4: fp4 x fp4, 2: fp6 x fp6
  const int l = threadIdx.x & 63;
  v16f acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const v8i av = a[l], bv = b[l];
  for (int it = 0; it < iters; it++) {
    acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc0, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc1, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc2, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    acc3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc3, F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  if (d) d[threadIdx.x + blockIdx.x * blockDim.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
__global__ void k_time(const v8i* a, const v8i* b, float* d, int iters) {
  const int l = threadIdx.x & 63;
  v16f acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const v8i av = a[l], bv = b[l];
  for (int it = 0; it < iters; it++) {
    acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc0, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc1, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc2, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    acc3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc3, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  if (d) d[threadIdx.x + blockIdx.x * blockDim.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k_time_i8(const v4i* a, const v4i* b, int* d, int iters) {
  const int l = threadIdx.x & 63;
  v16i acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const v4i av = a[l], bv = b[l];
  for (int it = 0; it < iters; it++) {
    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc3, 0, 0, 0);
  }
  if (d) d[threadIdx.x + blockIdx.x * blockDim.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

static float e4m3(uint8_t v) {   // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -x : x;
}
static float e5m2(uint8_t v) {
  const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  const float x = e == 0 ? ldexpf((float)m / 4.f, -14) : ldexpf(1.f + (float)m / 4.f, e - 15);
  return s ? -x : x;
}

int main() {
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  v8i *da, *db; float* dd;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 64 * 16 * 4);
  int bad_total = 0;
  {   // mode 2: fp4 nibbles 0x2 (1.0) for set bits on both sides, 32 of them in the low 16 bytes of a lane; B's block scale 2^7; C[i] = 100 i
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 32; j++) {
        A[l * 32 + j] = j < 16 ? (uint8_t)(((rnd() & 1) ? 0x02 : 0) | ((rnd() & 1) ? 0x20 : 0)) : (uint8_t)rnd();   // bytes 16..31: garbage the instruction must ignore
        B[l * 32 + j] = j < 16 ? (uint8_t)(((rnd() & 1) ? 0x02 : 0) | ((rnd() & 1) ? 0x20 : 0)) : (uint8_t)rnd();
      }
    hipMemcpy(da, A.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 64 * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd, 2);
    std::vector<float> D(64 * 16);
    hipMemcpy(D.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
        int common = 0;
        for (int h = 0; h < 2; h++)
          for (int j = 0; j < 16; j++) {
            const uint8_t a = A[(32 * h + row) * 32 + j], b = B[(32 * h + col) * 32 + j];
            common += ((a & b) >> 1 & 1) + ((a & b) >> 5 & 1);
          }
        const double ref = 128.0 * common + 100.0 * i;
        if ((double)D[l * 16 + i] != ref) { if (bad < 4) printf("mode 2 lane %d reg %d: got %.10g want %.10g\n", l, i, D[l * 16 + i], ref); bad++; }
      }
    printf("mode 2 (A, B fp4 0x2, scale_b 2^7, C = 100 i): %d of 1024 differ\n", bad);
    bad_total += bad;
  }
  for (int mode = 1; mode >= 0; mode--) {
    // mode 1: A bits -> 0x04 (bf8 2^-14), B bits -> 0x58 (bf8 2^7).  mode 0: A = e4m3 codes of small signed integers, B = bf8 0x04 / 0x1C / 0
    const uint8_t ints[] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50, 0x58, 0x5C, 0x60, 0xB8, 0xC8, 0xE0, 0xF0};   // 0..7, 8, 16, 24, 32, -1, -4, -32, -128
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 32; j++) {
        A[l * 32 + j] = mode ? ((rnd() & 1) ? 0x04 : 0x00) : ints[rnd() & 15];
        B[l * 32 + j] = mode ? ((rnd() & 1) ? 0x58 : 0x00) : (uint8_t[]){0x00, 0x04, 0x1C, 0x04}[rnd() & 3];
      }
    hipMemcpy(da, A.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 64 * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd, mode);
    std::vector<float> D(64 * 16);
    hipMemcpy(D.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
        double ref = 0;
        for (int h = 0; h < 2; h++)
          for (int j = 0; j < 32; j++) {
            const uint8_t a = A[(32 * h + row) * 32 + j], b = B[(32 * h + col) * 32 + j];
            ref += (double)(mode ? e5m2(a) : e4m3(a)) * (double)e5m2(b);
          }
        if ((double)D[l * 16 + i] != ref) { if (bad < 4) printf("mode %d lane %d reg %d: got %.10g want %.10g\n", mode, l, i, D[l * 16 + i], ref); bad++; }
      }
    printf("mode %d (%s): %d of 1024 differ\n", mode, mode ? "A bf8 0x04, B bf8 0x58" : "A fp8 e4m3 ints, B bf8 2^-14 / 2^-8", bad);
    bad_total += bad;
  }
  // timing: 4 independent accumulator chains per wavefront, one wavefront per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  float* dt; hipMalloc(&dt, 1024 * 256 * 4);
  for (int which = 0; which < 4; which++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, 0);
      if (which == 0) hipLaunchKernelGGL(k_time, dim3(256), dim3(256), 0, 0, da, db, dt, iters);
      else if (which == 2) hipLaunchKernelGGL(k_time_f<4>, dim3(256), dim3(256), 0, 0, da, db, dt, iters);
      else if (which == 3) hipLaunchKernelGGL(k_time_f<2>, dim3(256), dim3(256), 0, 0, da, db, dt, iters);
      else hipLaunchKernelGGL(k_time_i8, dim3(256), dim3(256), 0, 0, (const v4i*)da, (const v4i*)db, (int*)dt, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.3f ms for %d x 4 MFMA per wavefront (1 wavefront per SIMD) = %.1f ns per MFMA = %.1f cycles at 2.4 GHz\n",
                      which == 1 ? "v_mfma_i32_32x32x32_i8" : which == 2 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4)" : which == 3 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 x fp6)" : "v_mfma_scale_f32_32x32x64_f8f6f4 (bf8 x bf8)", ms, iters, ms * 1e6 / (iters * 4.0), ms * 1e6 / (iters * 4.0) * 2.4);
    }
  }
  return bad_total ? 1 : 0;
}
