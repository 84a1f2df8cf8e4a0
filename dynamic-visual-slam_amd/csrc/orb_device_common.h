// orb_device_common.h — device helpers shared by the two extractors of the library: orb_kernels.h (ORB_SLAM3::ORBextractor, the
// frontend's) and cvorb.hip (cv::ORB, row N4): the wavefront-level fence, cv::fastAtan2, the rBRIEF pattern, REFLECT_101 and the
// wavefront-parallel statement of libstdc++'s __unguarded_partition_pivot (lsort.h explains why it is exact).
#pragma once
#include <hip/hip_runtime.h>
#include "lsort.h"

namespace dvs {

// LDS / global operations of ONE wavefront execute in issue order; phases of wave-private algorithms are separated by this
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// rBRIEF sampling pattern bit_pattern_31_ (ORBextractor.cpp:149-407 = OpenCV's orb.cpp): 256 pairs as 1024 int8
static __constant__ int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};

// cv::fastAtan2 (atan_f32), float32 with individually rounded operations
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// __unguarded_partition_pivot(a + f, a + l) by one wavefront: median-of-3 to a[f], then the Hoare swaps as rank pairs (lsort.h:
// ranked_partition).  Elements are 64-bit words ordered by (word >> SHIFT); Lp / Rp: scratch of (l - f) ints each, indexed from f.
template <int SHIFT>
__device__ __forceinline__ int wave_partition(unsigned long long* a, int* Lp, int* Rp, int f, int l, int lane) {
  const lsort::Less<SHIFT> less;
  if (lane == 0) lsort::move_median_to_first(a + f, a + f + 1, a + f + (l - f) / 2, a + l - 1, less);
  wave_lds_fence();
  const unsigned long long pk = a[f] >> SHIFT;
  const unsigned long long ltm = (1ull << lane) - 1ull;
  int nge = 0, nle = 0;
  for (int c = f + 1; c < l; c += 64) {
    const int i = c + lane;
    const bool v = i < l;
    const unsigned long long k = v ? a[i] >> SHIFT : 0ull;
    const bool ge = v && !(k < pk), le = v && !(pk < k);
    const unsigned long long mg = __ballot(ge), ml = __ballot(le);
    if (ge) Lp[f + nge + __popcll(mg & ltm)] = i;
    if (le) Rp[f + nle + __popcll(ml & ltm)] = i;  // ascending; k-th from the right = Rp[f + nle - 1 - k]
    nge += __popcll(mg); nle += __popcll(ml);
  }
  wave_lds_fence();
  const int mm = min(nge, nle);
  int K = 0;
  for (int c = 0; c < mm; c += 64) {
    const int k = c + lane;
    const unsigned long long mk = __ballot(k < mm && Lp[f + k] < Rp[f + nle - 1 - k]);
    K += __popcll(mk);
    if (mk != ~0ull) break;  // the pairs that swap are a prefix
  }
  for (int c = 0; c < K; c += 64) {
    const int k = c + lane;
    if (k < K) {
      const int i = Lp[f + k], j = Rp[f + nle - 1 - k];
      const unsigned long long x = a[i], y = a[j];
      a[i] = y; a[j] = x;
    }
  }
  const int big = 0x7fffffff;
  const int lK = K < nge ? Lp[f + K] : big;
  const int rprev = K > 0 ? Rp[f + nle - K] : big;
  wave_lds_fence();
  return min(lK, rprev);
}

}  // namespace dvs
