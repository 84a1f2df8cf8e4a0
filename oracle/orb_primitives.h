// ============================================================================================
// oracle/orb_primitives.h — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp's header): the OpenCV / reference
// primitives shared by the two CPU restatements that need them — orb_oracle.cpp (ORB_SLAM3::ORBextractor as the
// frontend uses it) and cvorb_oracle.cpp (cv::ORB as test_dbow2_integration.cpp:19,38 uses it):
//   cv::resize INTER_LINEAR 8UC1, cv::FAST TYPE_9_16 + nonmax, cv::fastAtan2, IC_Angle, the steered
//   BRIEF descriptor, cv::GaussianBlur 7x7 fixed point.  Every definition sits in an unnamed namespace.
// ============================================================================================
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

using u8 = uint8_t;

// ---- OpenCV scalar helpers (fast_math.hpp) ---------------------------------------------------
inline int cvRoundF(float v) { return (int)lrintf(v); }    // round-half-even (default FP mode)
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
inline int cvCeilF(float v) { int i = (int)v; return i + (i < v); }
inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
inline short satShort(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

struct KeyPoint {  // same field order as cv::KeyPoint (28 bytes)
  float x, y, size, angle, response;
  int octave, class_id;
};

const int PATCH_SIZE = 31;       // ORBX.cpp:71
const int HALF_PATCH_SIZE = 15;  // ORBX.cpp:72
const int EDGE_THRESHOLD = 19;   // ORBX.cpp:73

const int8_t kPattern[1024] = {
#include "brief_pattern.inc"
};

struct Image {  // owning 8UC1 image, step == cols
  int cols = 0, rows = 0;
  std::vector<u8> d;
  void create(int c, int r) { cols = c; rows = r; d.assign((size_t)c * r, 0); }
  const u8* row(int y) const { return d.data() + (size_t)y * cols; }
  u8* row(int y) { return d.data() + (size_t)y * cols; }
};

// ---- cv::resize(src,dst,dsize,0,0,INTER_LINEAR), 8UC1, non-IPP (resize.cpp) ----------------------
// Called at ORBX.cpp:1182 with dsize = level size, src = previous level.
void resizeLinearU8(const u8* src, int sw, int sh, size_t sstep, u8* dst, int dw, int dh, size_t dstep) {
  const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * dw), beta(2 * dh);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloorF(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    alpha[2 * dx] = satShort(cvRoundF((1.f - fx) * COEF_SCALE));
    alpha[2 * dx + 1] = satShort(cvRoundF(fx * COEF_SCALE));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloorF(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = satShort(cvRoundF((1.f - fy) * COEF_SCALE));
    beta[2 * dy + 1] = satShort(cvRoundF(fy * COEF_SCALE));
  }
  std::vector<int> r0(dw), r1(dw);
  int have0 = -1, have1 = -1;  // source rows currently held in r0 / r1 (resize.cpp keeps its row buffers the same way)
  auto hline = [&](int sy, std::vector<int>& D) {
    const u8* S = src + (size_t)sy * sstep;
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      D[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
    }
    for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * COEF_SCALE;
  };
  for (int dy = 0; dy < dh; dy++) {
    const int s0 = std::min(std::max(yofs[dy], 0), sh - 1), s1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);  // clip(sy, 0, ssize.height)
    if (have0 != s0) {
      if (have1 == s0) { r0.swap(r1); std::swap(have0, have1); }
      else { hline(s0, r0); have0 = s0; }
    }
    if (have1 != s1) { hline(s1, r1); have1 = s1; }
    int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    u8* D = dst + (size_t)dy * dstep;
    const int* R0 = r0.data(); const int* R1 = r1.data();
    for (int x = 0; x < dw; x++)
      D[x] = (u8)((((b0 * (R0[x] >> 4)) >> 16) + ((b1 * (R1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---- cv::FAST(img, kps, threshold, true)  (fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) ---
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int cornerScore16(const u8* ptr, const int pixel[25], int threshold) {
  const int N = 25;
  int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

struct FastPt { int x, y, score; };

// img = sub-image (cols x rows, row stride `step`); output in sub-image coordinates, row-major.
void fast9_16(const u8* img, int cols, int rows, size_t step, int threshold, std::vector<FastPt>& out) {
  out.clear();
  if (cols < 7 || rows < 7) return;  // loops below never detect anything
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kRing[k][0] + kRing[k][1] * (int)step;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  // score buffer for the whole sub-image (the reference keeps a 3-row ring; same values)
  static thread_local std::vector<u8> sc, iscorner;
  sc.assign((size_t)cols * rows, 0);
  iscorner.assign((size_t)cols * rows, 0);
  // threshold_tab of fast.cpp: 1 = darker than v - t, 2 = brighter than v + t
  u8 threshold_tab[512];
  for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (u8)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  for (int i = 3; i < rows - 3; i++) {
    const u8* ptr = img + (size_t)i * step + 3;
    for (int j = 3; j < cols - 3; j++, ptr++) {
      int v = ptr[0];
      // high-speed rejection exactly as FAST_t<16> does it: every opposite pair needs a darker (brighter) sample
      const u8* tab = &threshold_tab[0] - v + 255;
      int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
      if (d == 0) continue;
      d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
      d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
      d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
      if (d == 0) continue;
      d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
      d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
      d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
      d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
      bool corner = false;
      if (d & 1) {  // darker arc: x < v - threshold
        int vt = v - threshold, count = 0;
        for (int k = 0; k < N; k++) {
          int x = ptr[pixel[k]];
          if (x < vt) { if (++count > K) { corner = true; break; } }
          else count = 0;
        }
      }
      if (!corner && (d & 2)) {  // brighter arc
        int vt = v + threshold, count = 0;
        for (int k = 0; k < N; k++) {
          int x = ptr[pixel[k]];
          if (x > vt) { if (++count > K) { corner = true; break; } }
          else count = 0;
        }
      }
      if (corner) {
        iscorner[(size_t)i * cols + j] = 1;
        sc[(size_t)i * cols + j] = (u8)cornerScore16(ptr, pixel, threshold);
      }
    }
  }
  for (int i = 3; i < rows - 3; i++)
    for (int j = 3; j < cols - 3; j++) {
      if (!iscorner[(size_t)i * cols + j]) continue;
      const u8* c = &sc[(size_t)i * cols + j];
      int s = c[0];
      if (s > c[-1] && s > c[1] && s > c[-cols - 1] && s > c[-cols] && s > c[-cols + 1] &&
          s > c[cols - 1] && s > c[cols] && s > c[cols + 1])
        out.push_back({j, i, s});
    }
}

// ---- cv::fastAtan2 (mathfuncs_core: atan_f32), float32, no contraction -----------------------
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
float fastAtan2f(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- IC_Angle  (ORBX.cpp:76-103) ------------------------------------------------------------
float icAngle(const u8* img, size_t stepB, float ptx, float pty, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const u8* center = img + (size_t)cvRoundF(pty) * stepB + cvRoundF(ptx);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  int step = (int)stepB;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fastAtan2f((float)m_01, (float)m_10);
}

// ---- computeOrbDescriptor (ORBX.cpp:106-146) ------------------------------------------------
const float factorPI = (float)(M_PI / 180.f);
void orbDescriptor(float kx, float ky, float kangle, const u8* img, size_t stepB, u8* desc) {
  float angle = (float)kangle * factorPI;
  float a = (float)cosf(angle), b = (float)sinf(angle);  // float overloads of cos/sin -> glibc cosf/sinf
  const u8* center = img + (size_t)cvRoundF(ky) * stepB + cvRoundF(kx);
  const int step = (int)stepB;
  const int8_t* pat = kPattern;
  auto val = [&](int idx) -> int {
    int px = pat[2 * idx], py = pat[2 * idx + 1];
    return center[cvRoundF(px * b + py * a) * step + cvRoundF(px * a - py * b)];
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int v = 0;
    for (int k = 0; k < 8; k++) {
      int t0 = val(2 * k), t1 = val(2 * k + 1);
      v |= (t0 < t1) << k;
    }
    desc[i] = (u8)v;
  }
}
#pragma GCC pop_options

// ---- cv::GaussianBlur(src,dst,Size(7,7),2,2,BORDER_REFLECT_101) 8UC1 fixed-point path ----------
// kernel k[7] is Q8 (sum 256): horizontal Q8.8 in u16, vertical Q16.16, round-half-up.
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
  return p;
}
void gaussBlur7(const Image& src, Image& dst, const int k[7]) {
  dst.create(src.cols, src.rows);
  const int W = src.cols, H = src.rows;
  std::vector<uint16_t> h((size_t)W * H);
  for (int y = 0; y < H; y++) {
    const u8* s = src.row(y);
    uint16_t* hr = &h[(size_t)y * W];
    for (int x = 0; x < W; x++) {
      unsigned acc = 0;
      if (x >= 3 && x + 3 < W) {  // interior: no border arithmetic
        const u8* p = s + x - 3;
        acc = k[0] * p[0] + k[1] * p[1] + k[2] * p[2] + k[3] * p[3] + k[4] * p[4] + k[5] * p[5] + k[6] * p[6];
      } else {
        for (int i = 0; i < 7; i++) acc += (unsigned)k[i] * s[reflect101(x + i - 3, W)];
      }
      hr[x] = (uint16_t)acc;
    }
  }
  for (int y = 0; y < H; y++) {
    u8* d = dst.row(y);
    const uint16_t* r[7];
    for (int j = 0; j < 7; j++) r[j] = &h[(size_t)reflect101(y + j - 3, H) * W];
    for (int x = 0; x < W; x++) {
      const uint32_t acc = (uint32_t)k[0] * r[0][x] + (uint32_t)k[1] * r[1][x] + (uint32_t)k[2] * r[2][x] + (uint32_t)k[3] * r[3][x] +
                           (uint32_t)k[4] * r[4][x] + (uint32_t)k[5] * r[5][x] + (uint32_t)k[6] * r[6][x];
      d[x] = (u8)((acc + 32768u) >> 16);
    }
  }
}

}  // namespace
