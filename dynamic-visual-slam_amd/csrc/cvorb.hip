// cvorb.hip — cv::ORB-compatible extractor behind the C-ABI: dvs_cvorb_*.
// Replaces cv::ORB::create(nfeatures, ...)->detectAndCompute(image, noArray(), keypoints, descriptors) as the reference calls it at
// /root/reference/dynamic_visual_slam/test/test_dbow2_integration.cpp:19,38.  OpenCV is not vendored by the reference; the
// algorithm is the published one of OpenCV 4.x (features2d/src/orb.cpp, keypoint.cpp; imgproc/src/resize.cpp), restated on the CPU
// in oracle/cvorb_oracle.cpp, which the -m gpu tests compare this file with bit for bit:
//   pyramid   every level from the previous one with INTER_LINEAR_EXACT (Q8.8 coefficients from double arithmetic)   k_cv_resize
//   FAST      TYPE_9_16, threshold 20, non-max suppression over the WHOLE level (not per cell)                         k_cv_fast, k_cv_nms_*
//   culling   runByImageBorder(31), retainBest(2 N_l) by FAST score, HARRIS response (7 x 7, k = 0.04), retainBest(N_l) k_cv_retain
//             — retainBest is std::nth_element + std::partition: the order they leave IS the output order (csrc/lsort.h)
//   angle     intensity centroid on the un-blurred level, cv::fastAtan2                                                k_cv_describe
//   rBRIEF    on the 7 x 7 Gaussian-blurred level, steered by cosf / sinf of the angle                                k_cv_blur, k_cv_describe
// This row is not on the reference's live path (lowest priority): the kernels are plain — one wavefront per level walks
// the selection — and are not tuned.  There is no CPU fallback.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>
#include "common.h"
#ifdef DVS_TEST_HOOKS
#include "../../include/dvslam_hip_test.h"
#endif
#include "glibc_sincosf.h"
#include "orb_device_common.h"

namespace dvs {
namespace {

typedef uint8_t u8;
constexpr int kCvMaxLevels = 16;
constexpr int kCvHalfPatch = 15, kCvPatch = 31, kCvHarrisBlock = 7;

struct CvLevel {
  int w, h, pitch;
  uint64_t off;        // bytes into the pyramid / blurred / score blocks
  int quota;           // nfeaturesPerLevel
  int keyOff, keyCap;  // slice of the key / scratch arrays
  int rowOff;          // slice of the per-row count array
  float scale;         // layerScale
  int tx, ty;          // slices of the axis tables (entries)
  int xlo, xhi, ylo, yhi;
};
struct CvGeom {
  int nlevels, edge, fastTh, harris;   // harris: 1 = HARRIS_SCORE, 0 = FAST_SCORE
  int gk[7];
  int umax[kCvHalfPatch + 2];
  CvLevel lv[kCvMaxLevels];
};

__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }

// cv::resize INTER_LINEAR_EXACT, 8UC1: Q8.8 horizontal sums in 16 bits, Q16 vertical sum, round half up (resize.cpp: hlineResize /
// vlineResize on ufixedpoint16).  Tables: ofs + packed (c0 | c1 << 16), built on the host in double (build_axis)
__global__ __launch_bounds__(256) void k_cv_resize(const u8* __restrict__ src, int sw, int sh, int sp, u8* __restrict__ dst, int dw, int dh, int dp,
                                                   const int* __restrict__ xofs, const uint32_t* __restrict__ xc, const int* __restrict__ yofs,
                                                   const uint32_t* __restrict__ yc, int xlo, int xhi, int ylo, int yhi) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  auto hsum = [&](int sy) -> uint32_t {
    const u8* S = src + (uint64_t)sy * sp;
    if (x < xlo) return (uint32_t)S[0] << 8;
    if (x >= xhi) return (uint32_t)S[sw - 1] << 8;
    const uint32_t c = xc[x];
    const int o = xofs[x];
    return ((c & 0xFFFFu) * S[o] + (c >> 16) * S[o + 1]) & 0xFFFFu;
  };
  u8 out;
  if (y < ylo || y >= yhi) {
    out = (u8)((hsum(y < ylo ? 0 : sh - 1) + 128u) >> 8);
  } else {
    const uint32_t b = yc[y];
    const int o = yofs[y];
    out = (u8)(((b & 0xFFFFu) * hsum(o) + (b >> 16) * hsum(o + 1) + 32768u) >> 16);
  }
  dst[(uint64_t)y * dp + x] = out;
}

// cornerScore<16> (fast_score.cpp) on the raw differences: the largest threshold for which the pixel is still a 9-of-16 corner
__device__ __forceinline__ int cv_corner_score(const u8* c, int P) {
  const int o[16] = {3 * P,      3 * P + 1,  2 * P + 2,  P + 3,  3,  -P + 3,  -2 * P + 2,  -3 * P + 1,
                     -3 * P,     -3 * P - 1, -2 * P - 2, -P - 3, -3, P - 3,   2 * P - 2,   3 * P - 1};
  const int v = c[0];
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = v - (int)c[o[k]];
  int lo3[16], hi3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lo3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    hi3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
  }
  int A = -1000, B = 1000;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    A = max(A, min(min(lo3[k], lo3[(k + 3) & 15]), lo3[(k + 6) & 15]));
    B = min(B, max(max(hi3[k], hi3[(k + 3) & 15]), hi3[(k + 6) & 15]));
  }
  return max(A, -B) - 1;
}

// FAST score map of every level: score where the pixel is a corner at the threshold (score >= threshold), 0 elsewhere and on the
// 3-pixel rim FAST never visits.  grid = (ceil(maxw / 256), maxh, nlevels)
__global__ __launch_bounds__(256) void k_cv_fast(CvGeom G, const u8* __restrict__ pyr, u8* __restrict__ score) {
  const CvLevel& L = G.lv[blockIdx.z];
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= L.w || y >= L.h) return;
  int s = 0;
  if (x >= 3 && y >= 3 && x < L.w - 3 && y < L.h - 3) {
    const u8* c = pyr + L.off + (uint64_t)y * L.pitch + x;
    const int P = L.pitch, v = c[0], t = G.fastTh;
    // fast.cpp's high-speed test: every opposite pair must hold a sample darker than v - t (or every pair one brighter than v + t)
    const int a0 = c[3 * P], a8 = c[-3 * P], a4 = c[3], a12 = c[-3];
    const bool dark = (a0 < v - t || a8 < v - t) && (a4 < v - t || a12 < v - t);
    const bool bright = (a0 > v + t || a8 > v + t) && (a4 > v + t || a12 > v + t);
    if (dark || bright) {
      const int sc = cv_corner_score(c, P);
      if (sc >= t) s = sc;
    }
  }
  score[L.off + (uint64_t)y * L.pitch + x] = (u8)s;
}

// non-max suppression (strict 3 x 3 maximum of the score map) restricted to runByImageBorder's rectangle [edge, w - edge) x
// [edge, h - edge); one workgroup per (row, level).  EMIT = 0: count per row; EMIT = 1: write the row's keys at rowBase[row] in x order
__device__ __forceinline__ uint32_t desc_order(float r) {   // greater response -> smaller word (response > is the comparison of retainBest)
  if (r == 0.f) r = 0.f;   // -0 -> +0
  const uint32_t u = __float_as_uint(r);
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}
template <int EMIT>
__global__ __launch_bounds__(256) void k_cv_nms(CvGeom G, const u8* __restrict__ score, int* __restrict__ rowCount, const int* __restrict__ rowBase,
                                                unsigned long long* __restrict__ keys) {
  __shared__ int wsum[4];
  const CvLevel& L = G.lv[blockIdx.y];
  const int e = G.edge, y = blockIdx.x;
  const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
  const bool live = y >= e && y < L.h - e && L.w > 2 * e && L.h > 2 * e && y < L.h;
  if (y >= L.h) return;
  int total = 0;
  if (live) {
    const u8* row = score + L.off + (uint64_t)y * L.pitch;
    const int P = L.pitch;
    for (int x0 = e; x0 < L.w - e; x0 += 256) {
      const int x = x0 + tid;
      bool keep = false;
      int s = 0;
      if (x < L.w - e) {
        const u8* c = row + x;
        s = c[0];
        keep = s > 0 && s > c[-1] && s > c[1] && s > c[-P - 1] && s > c[-P] && s > c[-P + 1] && s > c[P - 1] && s > c[P] && s > c[P + 1];
      }
      const unsigned long long m = __ballot(keep);
      if (ln == 0) wsum[wv] = __popcll(m);
      __syncthreads();
      int before = 0, all = 0;
      for (int k = 0; k < 4; k++) { if (k < wv) before += wsum[k]; all += wsum[k]; }
      if (EMIT && keep) {
        const int pos = rowBase[L.rowOff + y] + total + before + __popcll(m & ((1ull << ln) - 1ull));
        if (pos < L.keyCap) keys[L.keyOff + pos] = ((unsigned long long)desc_order((float)s) << 32) | ((unsigned)y << 16) | (unsigned)x;
      }
      total += all;
      __syncthreads();
    }
  }
  if (!EMIT && tid == 0) rowCount[L.rowOff + y] = total;
}

// exclusive scan of the per-row counts of one level (one workgroup per level; rows <= a few thousand)
__global__ __launch_bounds__(256) void k_cv_rowscan(CvGeom G, const int* __restrict__ rowCount, int* __restrict__ rowBase, int* __restrict__ levelCount) {
  __shared__ int part[256];
  const CvLevel& L = G.lv[blockIdx.x];
  const int tid = threadIdx.x;
  const int per = (L.h + 255) / 256, y0 = tid * per, y1 = min(L.h, y0 + per);
  int s = 0;
  for (int y = y0; y < y1; y++) s += rowCount[L.rowOff + y];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < 256; k++) { const int t = part[k]; part[k] = run; run += t; }
    levelCount[blockIdx.x] = min(run, L.keyCap);
  }
  __syncthreads();
  int run = part[tid];
  for (int y = y0; y < y1; y++) { rowBase[L.rowOff + y] = run; run += rowCount[L.rowOff + y]; }
}

// ---- KeyPointsFilter::retainBest by ONE wavefront on 64-bit keys (order word << 32 | y << 16 | x) ---------------------------------
__device__ void wave_nth_element(unsigned long long* a, int* Lp, int* Rp, int first, int nth, int last) {
  const lsort::Less<32> less;
  int depth = 0;
  for (int m = last - first; m > 1; m >>= 1) depth++;
  depth *= 2;
  while (last - first > 3) {
    if (depth == 0) {
      if (lane() == 0) { lsort::heap_select(a + first, a + nth + 1, a + last, less); lsort::swp(a + first, a + nth); }
      wave_lds_fence();
      return;
    }
    --depth;
    const int cut = wave_partition<32>(a, Lp, Rp, first, last, lane());
    if (cut <= nth) first = cut; else last = cut;
  }
  if (lane() == 0) lsort::insertion_sort(a + first, a + last, less);
  wave_lds_fence();
}

// std::partition(a + f, a + l, response >= ambiguous) = (word >> 32) <= amb: the k-th element from the left that fails is swapped with
// the k-th from the right that passes while the former lies left of the latter; returns the partition point
__device__ int wave_pred_partition(unsigned long long* a, int* Lp, int* Rp, int f, int l, uint32_t amb) {
  const int ln = lane();
  const unsigned long long ltm = (1ull << ln) - 1ull;
  int nf = 0, nt = 0;
  for (int c = f; c < l; c += 64) {
    const int i = c + ln;
    const bool v = i < l;
    const bool t = v && (uint32_t)(a[i] >> 32) <= amb;
    const bool fl = v && !t;
    const unsigned long long mf = __ballot(fl), mt = __ballot(t);
    if (fl) Lp[f + nf + __popcll(mf & ltm)] = i;
    if (t) Rp[f + nt + __popcll(mt & ltm)] = i;
    nf += __popcll(mf); nt += __popcll(mt);
  }
  wave_lds_fence();
  const int mm = min(nf, nt);
  int K = 0;
  for (int c = 0; c < mm; c += 64) {
    const int k = c + ln;
    const unsigned long long mk = __ballot(k < mm && Lp[f + k] < Rp[f + nt - 1 - k]);
    K += __popcll(mk);
    if (mk != ~0ull) break;
  }
  for (int c = 0; c < K; c += 64) {
    const int k = c + ln;
    if (k < K) {
      const int i = Lp[f + k], j = Rp[f + nt - 1 - k];
      const unsigned long long x = a[i], y = a[j];
      a[i] = y; a[j] = x;
    }
  }
  wave_lds_fence();
  return f + nt;
}

__device__ int wave_retain_best(unsigned long long* a, int n, int n_points, int* Lp, int* Rp) {
  if (!(n_points >= 0 && n > n_points)) return n;
  if (n_points == 0) return 0;
  wave_nth_element(a, Lp, Rp, 0, n_points - 1, n);
  const uint32_t amb = (uint32_t)(a[n_points - 1] >> 32);
  return wave_pred_partition(a, Lp, Rp, n_points, n, amb);
}

// one wavefront per level: retainBest(2 N) by FAST score, HARRIS responses of the survivors, retainBest(N) (orb.cpp computeKeyPoints)
__global__ __launch_bounds__(64) void k_cv_retain(CvGeom G, const u8* __restrict__ pyr, unsigned long long* __restrict__ keys, int* __restrict__ Lp,
                                                  int* __restrict__ Rp, const int* __restrict__ levelCount, int* __restrict__ finalCount) {
  const CvLevel& L = G.lv[blockIdx.x];
  unsigned long long* a = keys + L.keyOff;
  int* lp = Lp + L.keyOff; int* rp = Rp + L.keyOff;
  int n = levelCount[blockIdx.x];
  n = wave_retain_best(a, n, G.harris ? 2 * L.quota : L.quota, lp, rp);
  if (G.harris) {
    const float scale = __fdiv_rn(1.f, __fmul_rn((float)((1 << 2) * kCvHarrisBlock), 255.f));
    const float s4 = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
    const int P = L.pitch, r = kCvHarrisBlock / 2;
    for (int i = lane(); i < n; i += 64) {
      const uint32_t xy = (uint32_t)a[i];
      const int x0 = (int)(xy & 0xFFFFu), y0 = (int)(xy >> 16);
      const u8* p0 = pyr + L.off + (uint64_t)(y0 - r) * P + (x0 - r);
      int sa = 0, sb = 0, sc = 0;
      for (int wy = 0; wy < kCvHarrisBlock; wy++)
        for (int wx = 0; wx < kCvHarrisBlock; wx++) {
          const u8* p = p0 + wy * P + wx;
          const int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-P + 1] - (int)p[-P - 1]) + ((int)p[P + 1] - (int)p[P - 1]);
          const int Iy = ((int)p[P] - (int)p[-P]) * 2 + ((int)p[P - 1] - (int)p[-P - 1]) + ((int)p[P + 1] - (int)p[-P + 1]);
          sa += Ix * Ix; sb += Iy * Iy; sc += Ix * Iy;
        }
      const float fa = (float)sa, fb = (float)sb, fc = (float)sc;
      const float sum = __fadd_rn(fa, fb);
      const float resp = __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(0.04f, sum), sum)), s4);
      a[i] = ((unsigned long long)desc_order(resp) << 32) | xy;
    }
    wave_lds_fence();
    n = wave_retain_best(a, n, L.quota, lp, rp);
  }
  if (lane() == 0) finalCount[blockIdx.x] = n;
}

// cv::GaussianBlur(7 x 7, sigma 2, BORDER_REFLECT_101) fixed-point path: horizontal Q8.8, vertical Q16.16, round half up.  Plain.
__global__ __launch_bounds__(256) void k_cv_blur(CvGeom G, const u8* __restrict__ pyr, u8* __restrict__ blur) {
  const CvLevel& L = G.lv[blockIdx.z];
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= L.w || y >= L.h) return;
  int xs[7];
#pragma unroll
  for (int i = 0; i < 7; i++) xs[i] = reflect101(x + i - 3, L.w);
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const u8* row = pyr + L.off + (uint64_t)reflect101(y + j - 3, L.h) * L.pitch;
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) h += (uint32_t)G.gk[i] * row[xs[i]];
    acc += (uint32_t)G.gk[j] * (h & 0xFFFFu);
  }
  blur[L.off + (uint64_t)y * L.pitch + x] = (u8)((acc + 32768u) >> 16);
}

// ICAngles + computeOrbDescriptors + the final keypoint record; one wavefront per keypoint, levels in order (outBase = prefix of the
// per-level final counts, computed per wave: nlevels is tiny)
struct CvKeypoint { float x, y, size, angle, response; int32_t octave, class_id; };
__global__ __launch_bounds__(256) void k_cv_describe(CvGeom G, const u8* __restrict__ pyr, const u8* __restrict__ blur, const unsigned long long* __restrict__ keys,
                                                     const int* __restrict__ finalCount, CvKeypoint* __restrict__ outKp, u8* __restrict__ outDesc, int capacity,
                                                     int* __restrict__ nOut) {
  const int ln = lane();
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);   // wave = output row
  int level = -1, idx = 0, total = 0;
  for (int l = 0; l < G.nlevels; l++) {
    const int c = finalCount[l];
    if (level < 0 && gw < total + c) { level = l; idx = gw - total; }
    total += c;
  }
  if (gw == 0 && ln == 0) *nOut = total;
  if (level < 0 || gw >= capacity) return;
  const CvLevel& L = G.lv[level];
  const unsigned long long key = keys[L.keyOff + idx];
  const int x = (int)((uint32_t)key & 0xFFFFu), y = (int)(((uint32_t)key >> 16) & 0xFFFFu);
  const int P = L.pitch;
  // IC_Angle: rows v = -15 .. 15, lane = row (lanes 31.. idle), u in [-umax[|v|], umax[|v|]]
  int m10 = 0, m01 = 0;
  if (ln <= 2 * kCvHalfPatch) {
    const int v = ln - kCvHalfPatch, d = G.umax[v < 0 ? -v : v];
    const u8* c = pyr + L.off + (uint64_t)(y + v) * P + x;
    int su = 0, sm = 0;
    for (int u = -d; u <= d; u++) { const int I = c[u]; su += u * I; sm += I; }
    m10 = su; m01 = v * sm;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // keypoint in level-0 coordinates (pt *= scale), then the descriptor's centre back in the level: cvRound(pt * (1 / scale))
  const float px = __fmul_rn((float)x, L.scale), py = __fmul_rn((float)y, L.scale);
  const float inv = __fdiv_rn(1.f, L.scale);
  const int cx = __float2int_rn(__fmul_rn(px, inv)), cy = __float2int_rn(__fmul_rn(py, inv));
  const float arad = __fmul_rn(angle, (float)(3.14159265358979323846 / 180.f));
  const float a = gsc::cosf_(arad), b = gsc::sinf_(arad);
  const u8* center = blur + L.off + (uint64_t)cy * P + cx;
  unsigned long long words[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int pat = reinterpret_cast<const int*>(c_pattern)[64 * r + ln];   // pair 64 r + lane: (x0, y0, x1, y1) as int8
    const float x0 = (float)(int8_t)(pat & 0xff), y0 = (float)(int8_t)((pat >> 8) & 0xff);
    const float x1 = (float)(int8_t)((pat >> 16) & 0xff), y1 = (float)(int8_t)((pat >> 24) & 0xff);
    const int ix0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b))), iy0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
    const int ix1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b))), iy1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
    const int t0 = center[iy0 * P + ix0], t1 = center[iy1 * P + ix1];
    words[r] = __ballot(t0 < t1);
  }
  if (ln < 4) reinterpret_cast<unsigned long long*>(outDesc + (uint64_t)gw * 32)[ln] = ln == 0 ? words[0] : ln == 1 ? words[1] : ln == 2 ? words[2] : words[3];
  if (ln == 0) {
    // response: the HARRIS value (or the FAST score) is recovered from the key's order word
    const uint32_t asc = ~(uint32_t)(key >> 32);
    const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7FFFFFFFu) : ~asc;
    CvKeypoint kp;
    kp.x = px; kp.y = py; kp.size = __fmul_rn((float)kCvPatch, L.scale); kp.angle = angle; kp.response = __uint_as_float(u);
    kp.octave = level; kp.class_id = -1;
    outKp[gw] = kp;
  }
}

// test hook: wave_retain_best alone
#ifdef DVS_TEST_HOOKS
__global__ __launch_bounds__(64) void k_test_retain(unsigned long long* a, int n, int n_points, int* Lp, int* Rp, int* out_n) {
  const int r = wave_retain_best(a, n, n_points, Lp, Rp);
  if (threadIdx.x == 0) *out_n = r;
}
#endif

}  // namespace
}  // namespace dvs

using namespace dvs;

struct dvs_cvorb {
  dvs_cvorb_params prm;
  int device = 0;
  hipStream_t stream = nullptr;
  int rows = 0, cols = 0, capacity = 0;
  CvGeom G;
  uint64_t blockBytes = 0;
  int maxw = 0, maxh = 0, totalKeys = 0, totalRows = 0;
  u8 *d_pyr = nullptr, *d_blur = nullptr, *d_score = nullptr;
  int *d_xofs = nullptr, *d_yofs = nullptr; uint32_t *d_xc = nullptr, *d_yc = nullptr;
  int *d_rowCount = nullptr, *d_rowBase = nullptr, *d_levelCount = nullptr, *d_finalCount = nullptr, *d_Lp = nullptr, *d_Rp = nullptr, *d_nout = nullptr;
  unsigned long long* d_keys = nullptr;
  CvKeypoint* d_kps = nullptr; u8* d_desc = nullptr;
};

namespace {

inline int cv_round_f(float v) { return (int)lrintf(v); }
inline int cv_round_d(double v) { return (int)lrint(v); }
inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }

// interpolationLinear<uchar>::getcoeff over one axis (resize.cpp): softdouble arithmetic = IEEE double, one rounding per operation
// (this file is compiled -ffp-contract=off); coefficients are ufixedpoint16 = cvRound(fraction * 256)
void build_axis(int ssize, int dsize, std::vector<int>& ofs, std::vector<uint32_t>& coef, int& lo, int& hi) {
  lo = 0; hi = dsize;
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1.0 / inv_scale;
  for (int val = 0; val < dsize; val++) {
    const double fval = scale * ((double)val + 0.5) - 0.5;
    int ival = cv_floor_d(fval);
    uint32_t c0, c1;
    if (ival >= 0 && ssize > 1) {
      if (ival < ssize - 1) { c1 = (uint32_t)cv_round_d((fval - (double)ival) * 256.0); c0 = 256u - c1; }
      else { ival = ssize - 2; c0 = 0; c1 = 256; hi = std::min(hi, val); }
    } else { lo = std::max(lo, val + 1); ival = 0; c0 = 256; c1 = 0; }
    ofs.push_back(ival);
    coef.push_back(c0 | (c1 << 16));
  }
}

void cv_free(dvs_cvorb* h) {
  void* p[] = {h->d_pyr, h->d_blur, h->d_score, h->d_xofs, h->d_yofs, h->d_xc, h->d_yc, h->d_rowCount, h->d_rowBase, h->d_levelCount, h->d_finalCount,
               h->d_Lp, h->d_Rp, h->d_nout, h->d_keys, h->d_kps, h->d_desc};
  for (void* q : p) if (q) (void)hipFree(q);
  h->d_pyr = h->d_blur = h->d_score = nullptr; h->d_xofs = h->d_yofs = nullptr; h->d_xc = h->d_yc = nullptr;
  h->d_rowCount = h->d_rowBase = h->d_levelCount = h->d_finalCount = h->d_Lp = h->d_Rp = h->d_nout = nullptr;
  h->d_keys = nullptr; h->d_kps = nullptr; h->d_desc = nullptr; h->rows = h->cols = 0;
}

template <class T>
dvs_status up(T** d, const std::vector<T>& v) {
  DVS_HIP(hipMalloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) DVS_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return DVS_OK;
}

// level sizes, quotas, umax (orb.cpp: detectAndCompute / computeKeyPoints) and the device workspace for one resolution
dvs_status cv_prepare(dvs_cvorb* h, int rows, int cols, int capacity) {
  if (h->rows == rows && h->cols == cols && h->capacity >= capacity && h->d_pyr) return DVS_OK;
  DVS_HIP(hipStreamSynchronize(h->stream));
  cv_free(h);
  CvGeom& G = h->G;
  memset(&G, 0, sizeof(G));
  const dvs_cvorb_params& P = h->prm;
  const int nl = P.nlevels;
  G.nlevels = nl; G.edge = P.edge_threshold; G.fastTh = std::min(std::max(P.fast_threshold, 0), 255); G.harris = P.score_type == 0;
  const int gk[7] = {18, 34, 48, 56, 48, 34, 18};
  memcpy(G.gk, gk, sizeof(gk));
  {
    int v, v0;
    const int vmax = cv_floor_f(kCvHalfPatch * sqrtf(2.f) / 2 + 1), vmin = cv_ceil_f(kCvHalfPatch * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v) G.umax[v] = cv_round_d(sqrt((double)kCvHalfPatch * kCvHalfPatch - v * v));
    for (v = kCvHalfPatch, v0 = 0; v >= vmin; --v) {
      while (G.umax[v0] == G.umax[v0 + 1]) ++v0;
      G.umax[v] = v0;
      ++v0;
    }
  }
  const double scaleFactor = (double)P.scale_factor;   // ORB_Impl keeps the float argument in a double
  const float factor = (float)(1.0 / scaleFactor);
  float ndesired = P.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) { G.lv[l].quota = cv_round_f(ndesired); sum += G.lv[l].quota; ndesired *= factor; }
  G.lv[nl - 1].quota = std::max(P.nfeatures - sum, 0);
  std::vector<int> xofs, yofs; std::vector<uint32_t> xc, yc;
  uint64_t off = 0;
  int keyOff = 0, rowOff = 0;
  h->maxw = h->maxh = 0;
  for (int l = 0; l < nl; l++) {
    CvLevel& L = G.lv[l];
    L.scale = (float)pow(scaleFactor, (double)l);            // getScale(level, firstLevel = 0, scaleFactor)
    const float inv_scale = 1.0f / L.scale;
    L.w = cv_round_f(cols * inv_scale); L.h = cv_round_f(rows * inv_scale);
    if (L.w < 1 || L.h < 1) { set_error("level %d of a %d x %d image is empty", l, cols, rows); return DVS_ERR_UNSUPPORTED; }
    if (L.w > 65535 || L.h > 65535) { set_error("image too large"); return DVS_ERR_UNSUPPORTED; }
    L.pitch = (L.w + 63) & ~63;
    L.off = off; off += (uint64_t)L.pitch * L.h;
    off = (off + 255) & ~(uint64_t)255;
    const int e = G.edge, iw = std::max(L.w - 2 * e, 0), ih = std::max(L.h - 2 * e, 0);
    L.keyOff = keyOff; L.keyCap = ((iw + 1) / 2) * ((ih + 1) / 2) + 64;   // strict 3 x 3 maxima: at most one per 2 x 2 block
    keyOff += L.keyCap;
    L.rowOff = rowOff; rowOff += L.h;
    h->maxw = std::max(h->maxw, L.w); h->maxh = std::max(h->maxh, L.h);
    L.tx = (int)xofs.size(); L.ty = (int)yofs.size();
    if (l > 0) {
      build_axis(G.lv[l - 1].w, L.w, xofs, xc, L.xlo, L.xhi);
      build_axis(G.lv[l - 1].h, L.h, yofs, yc, L.ylo, L.yhi);
    }
  }
  h->blockBytes = off; h->totalKeys = keyOff; h->totalRows = rowOff;
  DVS_HIP(hipMalloc((void**)&h->d_pyr, off)); DVS_HIP(hipMalloc((void**)&h->d_blur, off)); DVS_HIP(hipMalloc((void**)&h->d_score, off));
  DVS_TRY(up(&h->d_xofs, xofs)); DVS_TRY(up(&h->d_yofs, yofs)); DVS_TRY(up(&h->d_xc, xc)); DVS_TRY(up(&h->d_yc, yc));
  DVS_HIP(hipMalloc((void**)&h->d_rowCount, (size_t)rowOff * 4)); DVS_HIP(hipMalloc((void**)&h->d_rowBase, (size_t)rowOff * 4));
  DVS_HIP(hipMalloc((void**)&h->d_levelCount, nl * 4)); DVS_HIP(hipMalloc((void**)&h->d_finalCount, nl * 4)); DVS_HIP(hipMalloc((void**)&h->d_nout, 4));
  DVS_HIP(hipMalloc((void**)&h->d_keys, (size_t)keyOff * 8)); DVS_HIP(hipMalloc((void**)&h->d_Lp, (size_t)keyOff * 4)); DVS_HIP(hipMalloc((void**)&h->d_Rp, (size_t)keyOff * 4));
  DVS_HIP(hipMalloc((void**)&h->d_kps, (size_t)std::max(capacity, 1) * sizeof(CvKeypoint))); DVS_HIP(hipMalloc((void**)&h->d_desc, (size_t)std::max(capacity, 1) * 32));
  h->rows = rows; h->cols = cols; h->capacity = capacity;
  return DVS_OK;
}

}  // namespace

extern "C" {

dvs_status dvs_cvorb_create(const dvs_cvorb_params* p, int32_t device, dvs_cvorb** out) {
  DVS_ARG(p && out);
  *out = nullptr;
  DVS_ARG(p->nfeatures >= 0 && p->scale_factor > 1.0f && p->nlevels >= 1 && p->nlevels <= kCvMaxLevels);
  DVS_ARG(p->score_type == 0 || p->score_type == 1);
  // the descriptor samples reach 18 px and the HARRIS block 4 px from a keypoint: OpenCV relies on its 32-pixel pyramid border for
  // smaller edge thresholds; this library does not keep one
  if (p->edge_threshold < 19) { set_error("edge_threshold %d < 19 is not supported (cv::ORB default: 31)", p->edge_threshold); return DVS_ERR_UNSUPPORTED; }
  if (p->first_level != 0 || p->wta_k != 2 || p->patch_size != 31) {
    set_error("only firstLevel = 0, WTA_K = 2, patchSize = 31 (cv::ORB's defaults, as the reference uses them) are built");
    return DVS_ERR_UNSUPPORTED;
  }
  DVS_TRY(check_device(device));
  dvs_cvorb* h = new (std::nothrow) dvs_cvorb();
  if (!h) { set_error("out of host memory"); return DVS_ERR_HIP; }
  h->prm = *p; h->device = device;
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; set_error("hipStreamCreate failed"); return DVS_ERR_HIP; }
  *out = h;
  return DVS_OK;
}

void dvs_cvorb_destroy(dvs_cvorb* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  cv_free(h);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

dvs_status dvs_cvorb_detect_and_compute(dvs_cvorb* h, const uint8_t* gray, int32_t rows, int32_t cols, size_t step, dvs_keypoint* kps, uint8_t* desc,
                                        int32_t capacity, int32_t* n_out) {
  DVS_ARG(h && n_out);
  *n_out = 0;
  if (!gray || rows <= 0 || cols <= 0) return DVS_OK;   // _image.empty(): detectAndCompute returns with nothing detected
  DVS_ARG(kps && desc && capacity >= 0 && step >= (size_t)cols);
  DVS_HIP(hipSetDevice(h->device));
  DVS_TRY(cv_prepare(h, rows, cols, capacity));
  const CvGeom& G = h->G;
  hipStream_t st = h->stream;
  DVS_HIP(hipMemcpy2DAsync(h->d_pyr + G.lv[0].off, G.lv[0].pitch, gray, step, cols, rows, hipMemcpyHostToDevice, st));
  for (int l = 1; l < G.nlevels; l++) {
    const CvLevel &S = G.lv[l - 1], &D = G.lv[l];
    hipLaunchKernelGGL(k_cv_resize, dim3((D.w + 255) / 256, D.h), dim3(256), 0, st, h->d_pyr + S.off, S.w, S.h, S.pitch, h->d_pyr + D.off, D.w, D.h, D.pitch,
                       h->d_xofs + D.tx, h->d_xc + D.tx, h->d_yofs + D.ty, h->d_yc + D.ty, D.xlo, D.xhi, D.ylo, D.yhi);
  }
  const dim3 pix((h->maxw + 255) / 256, h->maxh, G.nlevels);
  hipLaunchKernelGGL(k_cv_fast, pix, dim3(256), 0, st, G, h->d_pyr, h->d_score);
  hipLaunchKernelGGL(k_cv_nms<0>, dim3(h->maxh, G.nlevels), dim3(256), 0, st, G, h->d_score, h->d_rowCount, h->d_rowBase, h->d_keys);
  hipLaunchKernelGGL(k_cv_rowscan, dim3(G.nlevels), dim3(256), 0, st, G, h->d_rowCount, h->d_rowBase, h->d_levelCount);
  hipLaunchKernelGGL(k_cv_nms<1>, dim3(h->maxh, G.nlevels), dim3(256), 0, st, G, h->d_score, h->d_rowCount, h->d_rowBase, h->d_keys);
  hipLaunchKernelGGL(k_cv_retain, dim3(G.nlevels), dim3(64), 0, st, G, h->d_pyr, h->d_keys, h->d_Lp, h->d_Rp, h->d_levelCount, h->d_finalCount);
  hipLaunchKernelGGL(k_cv_blur, pix, dim3(256), 0, st, G, h->d_pyr, h->d_blur);
  // upper bound of the result without a round trip: the caller's capacity; rows past the count do nothing
  if (capacity > 0)
    hipLaunchKernelGGL(k_cv_describe, dim3((capacity + 3) / 4), dim3(256), 0, st, G, h->d_pyr, h->d_blur, h->d_keys, h->d_finalCount, (CvKeypoint*)h->d_kps,
                       h->d_desc, capacity, h->d_nout);
  DVS_HIP(hipGetLastError());
  int n = 0;
  if (capacity > 0) {
    DVS_HIP(hipMemcpyAsync(&n, h->d_nout, 4, hipMemcpyDeviceToHost, st));
  } else {   // count only
    int fc[kCvMaxLevels];
    DVS_HIP(hipMemcpyAsync(fc, h->d_finalCount, G.nlevels * 4, hipMemcpyDeviceToHost, st));
    DVS_HIP(hipStreamSynchronize(st));
    for (int l = 0; l < G.nlevels; l++) n += fc[l];
  }
  DVS_HIP(hipStreamSynchronize(st));
  *n_out = n;
  if (n > capacity) { set_error("%d keypoints > capacity %d (retainBest keeps every keypoint that ties with the last one)", n, capacity); return DVS_ERR_CAPACITY; }
  static_assert(sizeof(CvKeypoint) == sizeof(dvs_keypoint), "cv::KeyPoint layout");
  if (n) {
    DVS_HIP(hipMemcpy(kps, h->d_kps, (size_t)n * sizeof(dvs_keypoint), hipMemcpyDeviceToHost));
    DVS_HIP(hipMemcpy(desc, h->d_desc, (size_t)n * 32, hipMemcpyDeviceToHost));
  }
  return DVS_OK;
}

dvs_status dvs_cvorb_get_level(dvs_cvorb* h, int32_t level, int32_t blurred, uint8_t* dst, int32_t cap_bytes, int32_t* w, int32_t* hh) {
  DVS_ARG(h && dst && w && hh && h->d_pyr && level >= 0 && level < h->G.nlevels);
  const CvLevel& L = h->G.lv[level];
  *w = L.w; *hh = L.h;
  if ((int64_t)L.w * L.h > cap_bytes) return DVS_ERR_CAPACITY;
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  DVS_HIP(hipMemcpy2D(dst, L.w, (blurred ? h->d_blur : h->d_pyr) + L.off, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return DVS_OK;
}

#ifdef DVS_TEST_HOOKS   // libdvslam_hip_test.so only (include/dvslam_hip_test.h)
// test hook: KeyPointsFilter::retainBest on bare responses through the kernel's wavefront routine; perm[i] = original index
dvs_status dvs_test_retain_best_device(const float* responses, int32_t n, int32_t n_points, int32_t* perm, int32_t* n_kept) {
  DVS_ARG(n >= 0 && n_kept && (n == 0 || (responses && perm)));
  *n_kept = 0;
  if (n == 0) return DVS_OK;
  std::vector<unsigned long long> v(n);
  for (int i = 0; i < n; i++) {
    float r = responses[i]; if (r == 0.f) r = 0.f;
    uint32_t u; memcpy(&u, &r, 4);
    const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    v[i] = ((unsigned long long)(~asc) << 32) | (unsigned)i;
  }
  unsigned long long* d = nullptr; int *lp = nullptr, *rp = nullptr, *dn = nullptr;
  hipError_t e = hipMalloc(&d, 8 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&lp, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&rp, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&dn, 4);
  if (e == hipSuccess) e = hipMemcpy(d, v.data(), 8 * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_test_retain, dim3(1), dim3(64), 0, 0, d, n, n_points, lp, rp, dn); e = hipGetLastError(); }
  int kept = 0;
  if (e == hipSuccess) e = hipMemcpy(&kept, dn, 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(v.data(), d, 8 * (size_t)n, hipMemcpyDeviceToHost);
  (void)hipFree(d); (void)hipFree(lp); (void)hipFree(rp); (void)hipFree(dn);
  DVS_HIP(e);
  *n_kept = kept;
  for (int i = 0; i < kept; i++) perm[i] = (int)(uint32_t)v[i];
  return DVS_OK;
}
// the same through the sequential statement in lsort.h (host, no GPU): what the wavefront routine restates
void dvs_test_retain_best_host(const float* responses, int32_t n, int32_t n_points, int32_t* perm, int32_t* n_kept) {
  std::vector<unsigned long long> v(n);
  for (int i = 0; i < n; i++) {
    float r = responses[i]; if (r == 0.f) r = 0.f;
    uint32_t u; memcpy(&u, &r, 4);
    const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    v[i] = ((unsigned long long)(~asc) << 32) | (unsigned)i;
  }
  int kept = n;
  if (n_points >= 0 && n > n_points) {
    if (n_points == 0) kept = 0;
    else {
      lsort::nth_element(v.data(), v.data() + n_points - 1, v.data() + n, lsort::Less<32>());
      const uint32_t amb = (uint32_t)(v[n_points - 1] >> 32);
      unsigned long long* ne = lsort::partition(v.data() + n_points, v.data() + n, [amb](unsigned long long k) { return (uint32_t)(k >> 32) <= amb; });
      kept = (int)(ne - v.data());
    }
  }
  *n_kept = kept;
  for (int i = 0; i < kept; i++) perm[i] = (int)(uint32_t)v[i];
}

#endif  // DVS_TEST_HOOKS

}  // extern "C"
