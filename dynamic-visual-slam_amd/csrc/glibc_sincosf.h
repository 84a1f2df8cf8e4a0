// glibc_sincosf.h — device/host restatement of glibc (>= 2.28) sinf / cosf for |x| < 120, i.e. the
// algorithm of ARM optimized-routines sincosf (double-precision polynomial after a one-multiply
// range reduction; glibc sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h,s_sincosf_data.c}).
// The reference computes the BRIEF steering terms with (float)cos(angle), (float)sin(angle) on a
// float argument (ORBextractor.cpp:111-112), which resolves to glibc cosf/sinf on the host; a GPU
// cannot call glibc, and __cosf/ocml cosf differ in the last ulp, which flips cvRound() ties.
// tools/check_sincosf.c verifies this restatement bit-for-bit against the container's glibc 2.35
// for EVERY float in [0, 6.5] (1 087 373 313 values, 0 mismatches, with and without FMA).
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define GSC_HD __host__ __device__ __forceinline__
#else
#define GSC_HD inline
#endif

namespace gsc {

GSC_HD uint32_t asuint(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __float_as_uint(x);
#else
  uint32_t u; memcpy(&u, &x, 4); return u;
#endif
}
GSC_HD uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

struct Poly { double c0, c1, c2, c3, c4, s1, s2, s3; };

// n even: sine polynomial, n odd: cosine polynomial (sinf_poly in sincosf.h)
GSC_HD float sinf_poly(double x, double x2, bool negcos, int n) {
  const double sg = negcos ? -1.0 : 1.0;  // __sincosf_table[1] negates the cosine coefficients only
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    double x3 = x * x2;
    double s1 = s2c + x2 * s3c;
    double x7 = x3 * x2;
    double s = x + x3 * s1c;
    return (float)(s + x7 * s1);
  } else {
    const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5,
                 c3c = sg * -0x1.6c087e89a359dp-10, c4c = sg * 0x1.99343027bf8c3p-16;
    double x4 = x2 * x2;
    double c2 = c3c + x2 * c4c;
    double c1 = c0 + x2 * c1c;
    double x6 = x4 * x2;
    double c = c1 + x4 * c2c;
    return (float)(c + x6 * c2);
  }
}

GSC_HD double reduce_fast(double x, int* np) {
  double r = x * 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * 0x1.921FB54442D18p0;
}

// valid for |y| < 120 (the extractor only passes [0, 2*pi]); larger inputs are not handled
GSC_HD float sinf_(float y) {
  double x = y;
  int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return sinf_poly(x, x * x, false, 0);
  }
  x = reduce_fast(x, &n);
  double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  return sinf_poly(x * s, x * x, (n & 2) != 0, n);
}
GSC_HD float cosf_(float y) {
  double x = y;
  int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return sinf_poly(x, x * x, false, 1);
  }
  x = reduce_fast(x, &n);
  double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  return sinf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}

}  // namespace gsc
