// Pinned in / out for the small host entry points (match, RANSAC stages).  Shared by match.hip and ransac.hip.
#pragma once
#include <chrono>
#include <hip/hip_runtime.h>
#include "common.h"

namespace dvs {
// in / out of the host entry points without copy commands: inputs sit in the matcher's pinned block and k_io_import brings them to
// the device; k_io_export (one workgroup) writes the contiguous result region back into the pinned block and publishes a sequence
// number behind a system-scope fence, which the host polls (bounded spin, then the stream wait).
static __global__ __launch_bounds__(256) void k_io_import(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int ndw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ndw) dst[i] = src[i];
}
static __global__ __launch_bounds__(256) void k_io_export(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int ndw, int* __restrict__ hseq, int seq) {
  for (int i = threadIdx.x; i < ndw; i += 256) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { *reinterpret_cast<volatile int*>(hseq) = seq; __threadfence_system(); }
}
inline dvs_status io_wait(const volatile int* hseq, int seq, hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spin = 1; *hseq != seq; spin++) {
    __builtin_ia32_pause();
    if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (*hseq != seq) DVS_HIP(hipStreamSynchronize(st));
  return DVS_OK;
}

}  // namespace dvs
