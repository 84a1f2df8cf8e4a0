// frontend.hip — the glue either side of the hot path, so keypoints and descriptors can stay
// resident in HBM between extraction, matching and the backend's association:
//   dvs_bgr_to_gray*        cv::cvtColor(BGR2GRAY), 8-bit fixed point            frontend.cpp:1084
//   dvs_filter_depth*       filterDepth / isValidDepth (order-preserving compaction) frontend.cpp:457-527
//   dvs_filter_matches      distance < 50 filter                                 frontend.cpp:618-623, 1126-1132
//   dvs_backproject         publishKeyframe's depth back-projection               frontend.cpp:732-776
//   dvs_associate           associateObservation + reprojectPoint on a database snapshot  backend.cpp:1064-1173
// All of it is byte/integer or explicitly rounded float/double work: gathers and stream compactions, HBM/latency bound.
// The context is a dvs_matcher handle (stream + grow-only scratch), declared in match.hip.
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "io_pinned.h"

namespace dvs {

typedef unsigned long long u64;

// scratch + stream access to the matcher handle (match.hip)
dvs_status matcher_scratch(dvs_matcher* m, int slot, size_t bytes, void** out);
dvs_status matcher_pinned(dvs_matcher* m, size_t bytes, void** out, int** h_seq, int** counter);
hipStream_t matcher_stream(dvs_matcher* m);
int matcher_device(dvs_matcher* m);
// match.hip: candidate pairs (Hamming < max_dist) as (q, t, dist) triplets + per-query offsets, left on the device
dvs_status matcher_thresh_device(dvs_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int max_dist, const long long** d_offs,
                                 const int** d_pairs, long long* total);

__device__ __forceinline__ int blk_excl_scan(int v, int* wsum, int& total) {  // 256 threads
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if ((int)(threadIdx.x & 63) >= o) incl += t; }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { const int s = wsum[i]; if (i < w) base += s; }
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + incl - v;
}

// 4 pixels per thread; coefficients of OpenCV's RGB2Gray<uchar> (variant 0: 15-bit, 4.x; variant 1: 14-bit, older releases)
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* __restrict__ bgr, uint64_t step, uint64_t fstride, int rows, int cols,
                                                  uint8_t* __restrict__ gray, uint64_t gstep, uint64_t gfstride, int cb, int cg, int cr,
                                                  int shift) {
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y, f = blockIdx.z;
  if (x4 >= cols || y >= rows) return;
  const uint8_t* s = bgr + (uint64_t)f * fstride + (uint64_t)y * step + 3 * (uint64_t)x4;
  uint8_t* d = gray + (uint64_t)f * gfstride + (uint64_t)y * gstep + x4;
  const int rnd = 1 << (shift - 1);
  const int n = min(4, cols - x4);
  uint8_t px[12];
  if (n == 4 && (((uintptr_t)s) & 3) == 0) {
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s);
    const uint32_t a = s32[0], b = s32[1], c = s32[2];
    memcpy(px, &a, 4); memcpy(px + 4, &b, 4); memcpy(px + 8, &c, 4);
  } else {
    for (int i = 0; i < 3 * n; i++) px[i] = s[i];
  }
  uint32_t out = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < n) out |= (uint32_t)((px[3 * i] * cb + px[3 * i + 1] * cg + px[3 * i + 2] * cr + rnd) >> shift) << (8 * i);
  if (n == 4 && (((uintptr_t)d) & 3) == 0) *reinterpret_cast<uint32_t*>(d) = out;
  else for (int i = 0; i < n; i++) d[i] = (uint8_t)(out >> (8 * i));
}

// std::round(float): half away from zero
__device__ __forceinline__ int round_half_away(float v) { return (int)roundf(v); }

// one workgroup per frame: order-preserving compaction of keypoints / descriptors with a valid depth
__global__ __launch_bounds__(256) void k_filter_depth(const dvs_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                      const int* __restrict__ nArr, int nConst, int strideRows,
                                                      const uint16_t* __restrict__ depth, uint64_t dstep, uint64_t dfstride, int rows,
                                                      int cols, float dmin, float dmax, dvs_keypoint* __restrict__ okps,
                                                      uint8_t* __restrict__ odesc, int* __restrict__ oindex, int* __restrict__ nOut,
                                                      int* __restrict__ hseq, int seq) {
  __shared__ int wsum[5];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = nArr ? min(max(nArr[f], 0), strideRows) : nConst;  // a stale / corrupt count must not walk past the frame's block (as k_match)
  const dvs_keypoint* kp = kps + (size_t)f * strideRows;
  const uint8_t* dp = (const uint8_t*)depth + (uint64_t)f * dfstride;
  // Four chunks of 256 keypoints per trip with their loads in flight together — keypoints, then the depth pixels, then the
  // descriptors: through the host entry point every load is a PCIe round trip, and one chunk at a time made twelve of them per
  // thousand keypoints where three suffice.  The compaction order is unchanged.
  int carry = 0;
  for (int b = 0; b < n; b += 1024) {
    dvs_keypoint k[4];
    bool keep[4];
    uint4 d0[4], d1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = b + 256 * u + tid;
      if (i < n) k[u] = kp[i];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = b + 256 * u + tid;
      keep[u] = false;
      if (i < n) {
        const int x = round_half_away(k[u].x), y = round_half_away(k[u].y);
        if (x >= 0 && y >= 0 && x < cols && y < rows) {
          const float d = __fmul_rn((float)*(const uint16_t*)(dp + (uint64_t)y * dstep + 2 * (uint64_t)x), 0.001f);
          keep[u] = !(d < dmin || d > dmax);
        }
      }
    }
    if (desc) {
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (keep[u]) {
          const uint4* sd = reinterpret_cast<const uint4*>(desc + ((size_t)f * strideRows + b + 256 * u + tid) * 32);
          d0[u] = sd[0]; d1[u] = sd[1];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (b + 256 * u >= n) break;
      const int i = b + 256 * u + tid;
      int tot;
      const int pos = carry + blk_excl_scan(keep[u] ? 1 : 0, wsum, tot);
      if (keep[u]) {
        okps[(size_t)f * strideRows + pos] = k[u];
        if (oindex) oindex[(size_t)f * strideRows + pos] = i;
        if (desc) {
          uint4* dd = reinterpret_cast<uint4*>(odesc + ((size_t)f * strideRows + pos) * 32);
          dd[0] = d0[u]; dd[1] = d1[u];
        }
      }
      carry += tot;
    }
  }
  if (tid == 0) nOut[f] = carry;
  if (hseq) {   // host entry point (one workgroup, inputs and outputs in the pinned block): publish for the polling host
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { *reinterpret_cast<volatile int*>(hseq) = seq; __threadfence_system(); }
  }
}

__global__ __launch_bounds__(256) void k_filter_matches(const int* __restrict__ idx, const int* __restrict__ dist, const int* __restrict__ nArr,
                                                        int nConst, int strideRows, float maxd, int* __restrict__ out, int* __restrict__ nOut) {
  __shared__ int wsum[5];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = nArr ? min(max(nArr[f], 0), strideRows) : nConst;  // a stale / corrupt count must not walk past the frame's block (as k_match)
  int carry = 0;
  for (int b = 0; b < n; b += 256) {
    const int i = b + tid;
    const bool keep = i < n && (float)dist[(size_t)f * strideRows + i] < maxd;
    int tot;
    const int pos = carry + blk_excl_scan(keep ? 1 : 0, wsum, tot);
    if (keep) {
      int* o = out + 3 * ((size_t)f * strideRows + pos);
      o[0] = i; o[1] = idx[(size_t)f * strideRows + i]; o[2] = dist[(size_t)f * strideRows + i];
    }
    carry += tot;
  }
  if (tid == 0) nOut[f] = carry;
}

__global__ __launch_bounds__(256) void k_backproject(const dvs_keypoint* __restrict__ kps, int n, const uint16_t* __restrict__ depth,
                                                     int rows, int cols, uint64_t dstep, float fx, float fy, float cx, float cy, const double* __restrict__ Rt,
                                                     double* __restrict__ world, int* __restrict__ oindex, int* __restrict__ nOut) {
  __shared__ int wsum[5];
  const int tid = threadIdx.x;
  int carry = 0;
  for (int b = 0; b < n; b += 256) {
    const int i = b + tid;
    bool keep = false;
    float X = 0, Y = 0, Z = 0;
    if (i < n) {
      const float px = kps[i].x, py = kps[i].y;
      const int x = round_half_away(px), y = round_half_away(py);
      // a keypoint outside the depth image is undefined behaviour in the reference (cv::Mat::at, frontend.cpp:738); here it reads as
      // depth 0 and is dropped instead of faulting the GPU
      const bool inb = x >= 0 && y >= 0 && x < cols && y < rows;
      Z = inb ? __fmul_rn((float)*(const uint16_t*)((const uint8_t*)depth + (uint64_t)y * dstep + 2 * (uint64_t)x), 0.001f) : 0.f;
      X = __fdiv_rn(__fmul_rn(__fsub_rn(px, cx), Z), fx);
      Y = __fdiv_rn(__fmul_rn(__fsub_rn(py, cy), Z), fy);
      keep = (double)Z > 0.3 && (double)Z < 3.0;
    }
    int tot;
    const int pos = carry + blk_excl_scan(keep ? 1 : 0, wsum, tot);
    if (keep) {
      const double v0 = X, v1 = Y, v2 = Z;
#pragma unroll
      for (int r = 0; r < 3; r++) world[3 * (size_t)pos + r] = (Rt[3 * r] * v0 + Rt[3 * r + 1] * v1 + Rt[3 * r + 2] * v2) + Rt[9 + r];
      oindex[pos] = i;
    }
    carry += tot;
  }
  if (tid == 0) *nOut = carry;
}

// reprojectPoint (backend.cpp:1153-1173) + cv::norm for every candidate pair; Rt = R (9, row-major) followed by t (3)
__global__ __launch_bounds__(256) void k_reproject_errors(const int* __restrict__ pairs3, long long npairs, const float* __restrict__ obs_px,
                                                          const float* __restrict__ lm_xyz, const double* __restrict__ Rt, double fx,
                                                          double fy, double cx, double cy, double* __restrict__ err) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npairs) return;
  const float* o = obs_px + 2 * (size_t)pairs3[3 * p];
  const float* l = lm_xyz + 3 * (size_t)pairs3[3 * p + 1];
  const double d0 = (double)l[0] - Rt[9], d1 = (double)l[1] - Rt[10], d2 = (double)l[2] - Rt[11];
  const double c0 = Rt[0] * d0 + Rt[3] * d1 + Rt[6] * d2;
  const double c1 = Rt[1] * d0 + Rt[4] * d1 + Rt[7] * d2;
  const double c2 = Rt[2] * d0 + Rt[5] * d1 + Rt[8] * d2;
  float u = -1.f, v = -1.f;
  if (!(c2 <= 0)) { u = (float)(fx * c0 / c2 + cx); v = (float)(fy * c1 / c2 + cy); }
  const float dx = __fsub_rn(o[0], u), dy = __fsub_rn(o[1], v);
  err[p] = sqrt((double)dx * dx + (double)dy * dy);
}

// per observation: first candidate (landmark order) with the smallest error below the gate
__global__ __launch_bounds__(256) void k_assoc_argmin(const long long* __restrict__ offs, int nobs, const int* __restrict__ pairs3,
                                                      const double* __restrict__ err, double max_reproj, int* __restrict__ best) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nobs) return;
  int bl = -1;
  double be = DBL_MAX;
  for (long long p = offs[i]; p < offs[i + 1]; p++) {
    const double e = err[p];
    if (e < max_reproj && e < be) { bl = pairs3[3 * p + 1]; be = e; }
  }
  best[i] = bl;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Keyframe.msg on the wire: publishKeyframe (frontend.cpp:699-776) fused with the CDR serialisation
// rmw_fastrtps applies to dynamic_visual_slam_interfaces/msg/Keyframe (XCDR1, little endian, alignment relative to the byte
// after the 4-byte encapsulation header).  With 32-byte descriptors every array element has a fixed stride, so the payload is
// written in place by the threads that back-project the keypoints:
//   P+0   int32 stamp.sec | uint32 stamp.nanosec | uint32 len | header.frame_id chars + NUL
//   o1 = align8(12 + len): uint64 frame_id | translation x y z | rotation x y z w           (f64)
//   o2 = o1 + 64:          uint32 #landmarks | (pad to 8) | { uint64 landmark_id, x, y, z } * m              32 B each
//   o3 = o2 + 8 + 32 m:    uint32 #observations | (pad to 8) | { uint64 landmark_id, pixel_x, pixel_y (f64), uint32 32, desc[32] },
//                          stride 64 (the 4 pad bytes before the next uint64 are not written after the last element)
// m = 0: the two counts sit at o2 and o2 + 4 and the payload ends at o2 + 8 (padding is only emitted in front of an element).
// ---------------------------------------------------------------------------------------------------------------------------
struct KfHead { int32_t sec; uint32_t nanosec; uint32_t slen; uint32_t o1; unsigned long long keyframe_id; double pose[7]; char frame_id[64]; };

__device__ __forceinline__ void put_u64(uint8_t* p, unsigned long long v) {  // 4-byte aligned destinations
  reinterpret_cast<uint32_t*>(p)[0] = (uint32_t)v; reinterpret_cast<uint32_t*>(p)[1] = (uint32_t)(v >> 32);
}
__device__ __forceinline__ void put_f64(uint8_t* p, double v) { put_u64(p, (unsigned long long)__double_as_longlong(v)); }

__global__ __launch_bounds__(256) void k_publish_keyframe(KfHead H, const dvs_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                          int n, const uint16_t* __restrict__ depth, int rows, int cols, uint64_t dstep,
                                                          float fx, float fy,
                                                          float cx, float cy, const double* __restrict__ Rt, uint8_t* __restrict__ out,
                                                          unsigned long long cap, unsigned long long* __restrict__ outSize,
                                                          int* __restrict__ nOut) {
  __shared__ int wsum[5];
  __shared__ int s_m;
  const int tid = threadIdx.x;
  uint8_t* P = out + 4;
  const uint32_t o2 = H.o1 + 64;
  // pass 1: how many keypoints survive the depth gate (the observation array's offset depends on it)
  int cnt = 0;
  for (int i = tid; i < n; i += 256) {
    const int x = round_half_away(kps[i].x), y = round_half_away(kps[i].y);
    const bool inb = x >= 0 && y >= 0 && x < cols && y < rows;  // outside the depth image: dropped (undefined in the reference)
    const float Z = inb ? __fmul_rn((float)*(const uint16_t*)((const uint8_t*)depth + (uint64_t)y * dstep + 2 * (uint64_t)x), 0.001f) : 0.f;
    cnt += ((double)Z > 0.3 && (double)Z < 3.0) ? 1 : 0;
  }
  int tot;
  (void)blk_excl_scan(cnt, wsum, tot);
  if (tid == 0) s_m = tot;
  __syncthreads();
  const int m = s_m;
  const unsigned long long total = 4ull + (m == 0 ? o2 + 8ull : o2 + 8ull + 32ull * m + 8ull + 64ull * (m - 1) + 60ull);
  if (tid == 0) { *outSize = total; *nOut = m; }
  if (total > cap) return;  // the host reports DVS_ERR_CAPACITY from outSize
  if (tid == 0) {
    out[0] = 0; out[1] = 1; out[2] = 0; out[3] = 0;  // CDR_LE encapsulation, no options
    reinterpret_cast<int32_t*>(P)[0] = H.sec; reinterpret_cast<uint32_t*>(P)[1] = H.nanosec; reinterpret_cast<uint32_t*>(P)[2] = H.slen;
    for (uint32_t k = 0; k < H.slen; k++) P[12 + k] = (uint8_t)H.frame_id[k];
    for (uint32_t k = 12 + H.slen; k < H.o1; k++) P[k] = 0;  // alignment padding
    put_u64(P + H.o1, H.keyframe_id);
    for (int k = 0; k < 7; k++) put_f64(P + H.o1 + 8 + 8 * k, H.pose[k]);
    reinterpret_cast<uint32_t*>(P + o2)[0] = (uint32_t)m;
    if (m == 0) reinterpret_cast<uint32_t*>(P + o2)[1] = 0u;
    else {
      reinterpret_cast<uint32_t*>(P + o2)[1] = 0u;  // pad
      const uint32_t o3 = o2 + 8 + 32 * m;
      reinterpret_cast<uint32_t*>(P + o3)[0] = (uint32_t)m; reinterpret_cast<uint32_t*>(P + o3)[1] = 0u;
    }
  }
  if (m == 0) return;
  const uint32_t lm0 = o2 + 8, ob0 = o2 + 8 + 32 * m + 8;
  // pass 2: ordered compaction, landmark + observation records
  int carry = 0;
  for (int b = 0; b < n; b += 256) {
    const int i = b + tid;
    bool keep = false;
    float X = 0, Y = 0, Z = 0, px = 0, py = 0;
    if (i < n) {
      px = kps[i].x; py = kps[i].y;
      const int x = round_half_away(px), y = round_half_away(py);
      const bool inb = x >= 0 && y >= 0 && x < cols && y < rows;
      Z = inb ? __fmul_rn((float)*(const uint16_t*)((const uint8_t*)depth + (uint64_t)y * dstep + 2 * (uint64_t)x), 0.001f) : 0.f;
      X = __fdiv_rn(__fmul_rn(__fsub_rn(px, cx), Z), fx);
      Y = __fdiv_rn(__fmul_rn(__fsub_rn(py, cy), Z), fy);
      keep = (double)Z > 0.3 && (double)Z < 3.0;
    }
    int t2;
    const int pos = carry + blk_excl_scan(keep ? 1 : 0, wsum, t2);
    if (keep) {
      const double v0 = X, v1 = Y, v2 = Z;
      uint8_t* L = P + lm0 + 32 * (size_t)pos;
      put_u64(L, (unsigned long long)i);                                     // landmark_id = keypoint index (:758)
#pragma unroll
      for (int r = 0; r < 3; r++) put_f64(L + 8 + 8 * r, (Rt[3 * r] * v0 + Rt[3 * r + 1] * v1 + Rt[3 * r + 2] * v2) + Rt[9 + r]);
      uint8_t* O = P + ob0 + 64 * (size_t)pos;
      put_u64(O, (unsigned long long)i);
      put_f64(O + 8, (double)px); put_f64(O + 16, (double)py);               // float -> float64 fields (:764-765)
      reinterpret_cast<uint32_t*>(O + 24)[0] = 32u;
      const uint32_t* d = reinterpret_cast<const uint32_t*>(desc + 32 * (size_t)i);
#pragma unroll
      for (int k = 0; k < 8; k++) reinterpret_cast<uint32_t*>(O + 28)[k] = d[k];
      if (pos + 1 < m) reinterpret_cast<uint32_t*>(O + 60)[0] = 0u;          // pad in front of the next element
    }
    carry += t2;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Harris corner measure as cv::ORB scores its keypoints (HARRIS_SCORE, OpenCV features2d orb.cpp HarrisResponses; SURVEY.md §8f
// row N4): 3x3 Sobel-like integer gradients over a blockSize x blockSize window around the keypoint, a = sum Ix^2, b = sum Iy^2,
// c = sum Ix Iy (int32, exact), response = (a*b - c*c - k (a+b)^2) * scale^4 in float with scale = 1 / (4 blockSize 255).
// One wavefront per keypoint, one lane per window pixel (blockSize <= 8).  Points closer than blockSize/2 + 1 to the border
// have no defined response in OpenCV (it reads outside the layer); they get 0 here.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_harris(const uint8_t* __restrict__ img, int rows, int cols, uint64_t step, const int* __restrict__ xs,
                                                const int* __restrict__ ys, int n, int blockSize, float k, float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int r = blockSize / 2;
  const int x0 = xs[i], y0 = ys[i];
  const bool inside = x0 - r - 1 >= 0 && y0 - r - 1 >= 0 && x0 - r + blockSize <= cols - 1 && y0 - r + blockSize <= rows - 1;
  int a = 0, b = 0, c = 0;
  if (inside && lane < blockSize * blockSize) {
    const int wy = lane / blockSize, wx = lane - wy * blockSize;
    const uint8_t* p = img + (uint64_t)(y0 - r + wy) * step + (x0 - r + wx);
    const int64_t st = (int64_t)step;
    const int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-st + 1] - (int)p[-st - 1]) + ((int)p[st + 1] - (int)p[st - 1]);
    const int Iy = ((int)p[st] - (int)p[-st]) * 2 + ((int)p[st - 1] - (int)p[-st - 1]) + ((int)p[st + 1] - (int)p[-st + 1]);
    a = Ix * Ix; b = Iy * Iy; c = Ix * Iy;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  if (lane == 0) {
    float resp = 0.f;
    if (inside) {
      const float scale = __fdiv_rn(1.f, __fmul_rn((float)((1 << 2) * blockSize), 255.f));
      const float s4 = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
      const float fa = (float)a, fb = (float)b, fc = (float)c;
      const float sum = __fadd_rn(fa, fb);
      resp = __fmul_rn(__fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(k, sum), sum)), s4);
    }
    out[i] = resp;
  }
}
}  // namespace dvs

using namespace dvs;

extern "C" {

dvs_status dvs_bgr_to_gray_device(dvs_matcher* ctx, const uint8_t* d_bgr, int32_t nimg, int32_t rows, int32_t cols, size_t step,
                                  size_t frame_stride, uint8_t* d_gray, size_t gray_step, size_t gray_frame_stride, int32_t variant) {
  DVS_ARG(ctx && d_bgr && d_gray && nimg >= 0 && rows > 0 && cols > 0 && step >= (size_t)cols * 3 && gray_step >= (size_t)cols);
  DVS_ARG(variant == 0 || variant == 1);
  if (nimg == 0) return DVS_OK;
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  const int cb = variant == 0 ? 3735 : 1868, cg = variant == 0 ? 19235 : 9617, cr = variant == 0 ? 9798 : 4899, shift = variant == 0 ? 15 : 14;
  hipLaunchKernelGGL(k_bgr2gray, dim3((cols + 255) / 256, (rows + 3) / 4, nimg), dim3(64, 4), 0, matcher_stream(ctx), d_bgr, (uint64_t)step,
                     (uint64_t)frame_stride, rows, cols, d_gray, (uint64_t)gray_step, (uint64_t)gray_frame_stride, cb, cg, cr, shift);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status dvs_bgr_to_gray(dvs_matcher* ctx, const uint8_t* bgr, int32_t rows, int32_t cols, size_t step, uint8_t* gray, size_t gray_step,
                           int32_t variant) {
  DVS_ARG(ctx && bgr && gray && rows > 0 && cols > 0);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  void *d_in, *d_out;
  DVS_TRY(matcher_scratch(ctx, 0, (size_t)rows * cols * 3, &d_in));
  DVS_TRY(matcher_scratch(ctx, 1, (size_t)rows * cols, &d_out));
  hipStream_t st = matcher_stream(ctx);
  DVS_HIP(hipMemcpy2DAsync(d_in, (size_t)cols * 3, bgr, step, (size_t)cols * 3, rows, hipMemcpyHostToDevice, st));
  DVS_TRY(dvs_bgr_to_gray_device(ctx, (const uint8_t*)d_in, 1, rows, cols, (size_t)cols * 3, 0, (uint8_t*)d_out, cols, 0, variant));
  DVS_HIP(hipMemcpy2DAsync(gray, gray_step, d_out, cols, cols, rows, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipStreamSynchronize(st));
  return DVS_OK;
}

dvs_status dvs_filter_depth_batch_device(dvs_matcher* ctx, const dvs_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                         int32_t stride_rows, int32_t nframes, const uint16_t* d_depth, int32_t rows, int32_t cols,
                                         size_t step_bytes, size_t frame_stride_bytes, float min_depth, float max_depth,
                                         dvs_keypoint* d_out_kps, uint8_t* d_out_desc, int32_t* d_out_index, int32_t* d_n_out) {
  DVS_ARG(ctx && d_kps && d_n && d_depth && d_out_kps && d_n_out && nframes >= 0 && stride_rows > 0 && rows > 0 && cols > 0);
  DVS_ARG(!d_desc || d_out_desc);
  if (nframes == 0) return DVS_OK;
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipLaunchKernelGGL(k_filter_depth, dim3(nframes), dim3(256), 0, matcher_stream(ctx), d_kps, d_desc, d_n, 0, stride_rows, d_depth,
                     (uint64_t)step_bytes, (uint64_t)frame_stride_bytes, rows, cols, min_depth, max_depth, d_out_kps, d_out_desc, d_out_index, d_n_out,
                     (int*)nullptr, 0);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status dvs_filter_depth(dvs_matcher* ctx, const dvs_keypoint* kps, const uint8_t* desc, int32_t n, const uint16_t* depth, int32_t rows,
                            int32_t cols, size_t step_bytes, float min_depth, float max_depth, dvs_keypoint* out_kps, uint8_t* out_desc,
                            int32_t* out_index, int32_t* n_out) {
  DVS_ARG(ctx && n_out && n >= 0 && rows > 0 && cols > 0 && depth);
  *n_out = 0;
  if (n == 0) return DVS_OK;
  DVS_ARG(kps && out_kps && (!desc || out_desc));
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  // Everything through the matcher's pinned block, read and written by the kernel itself over PCIe (each byte once; of the depth
  // image only the n pixels under the keypoints): no copy commands, and the host polls the sequence number the kernel publishes.
  const size_t kb = ((size_t)n * sizeof(dvs_keypoint) + 15) & ~(size_t)15, db = (size_t)n * 32, ib = ((size_t)n * 4 + 16 + 15) & ~(size_t)15;
  const size_t zb = (size_t)rows * cols * 2;
  uint8_t* hio; int *hseq, *counter;
  DVS_TRY(matcher_pinned(ctx, 2 * kb + 2 * db + ib + zb, (void**)&hio, &hseq, &counter));
  dvs_keypoint* p_k = (dvs_keypoint*)hio; dvs_keypoint* p_ok = (dvs_keypoint*)(hio + kb);
  uint8_t* p_d = hio + 2 * kb; uint8_t* p_od = p_d + db;
  int* p_oi = (int*)(p_od + db); int* p_no = p_oi + n;
  uint16_t* p_dep = (uint16_t*)(hio + 2 * kb + 2 * db + ib);
  memcpy(p_k, kps, (size_t)n * sizeof(dvs_keypoint));
  if (desc) memcpy(p_d, desc, db);
  if (step_bytes == (size_t)cols * 2) memcpy(p_dep, depth, zb);
  else for (int r = 0; r < rows; r++) memcpy((uint8_t*)p_dep + (size_t)r * cols * 2, (const uint8_t*)depth + (size_t)r * step_bytes, (size_t)cols * 2);
  const int seq = ++*counter;
  hipLaunchKernelGGL(k_filter_depth, dim3(1), dim3(256), 0, st, p_k, desc ? p_d : nullptr, (const int*)nullptr, n, n, p_dep, (uint64_t)cols * 2,
                     (uint64_t)0, rows, cols, min_depth, max_depth, p_ok, p_od, p_oi, p_no, hseq, seq);
  DVS_HIP(hipGetLastError());
  DVS_TRY(io_wait(hseq, seq, st));
  const int m = *p_no;
  if (m) {
    memcpy(out_kps, p_ok, (size_t)m * sizeof(dvs_keypoint));
    if (desc) memcpy(out_desc, p_od, (size_t)m * 32);
    if (out_index) memcpy(out_index, p_oi, (size_t)m * 4);
  }
  *n_out = m;
  return DVS_OK;
}

dvs_status dvs_filter_matches(dvs_matcher* ctx, const int32_t* train_idx, const int32_t* dist, int32_t n, float max_distance,
                              int32_t* out_triplets, int32_t* n_out) {
  DVS_ARG(ctx && n_out && n >= 0);
  *n_out = 0;
  if (n == 0) return DVS_OK;
  DVS_ARG(train_idx && dist && out_triplets);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  int* base;
  DVS_TRY(matcher_scratch(ctx, 0, (size_t)n * 4 * 5 + 16, (void**)&base));
  int *d_i = base, *d_d = base + n, *d_o = base + 2 * n, *d_n = base + 5 * n;
  DVS_HIP(hipMemcpyAsync(d_i, train_idx, (size_t)n * 4, hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpyAsync(d_d, dist, (size_t)n * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_filter_matches, dim3(1), dim3(256), 0, st, d_i, d_d, (const int*)nullptr, n, n, max_distance, d_o, d_n);
  DVS_HIP(hipGetLastError());
  int m = 0;
  DVS_HIP(hipMemcpyAsync(&m, d_n, 4, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipStreamSynchronize(st));
  if (m) DVS_HIP(hipMemcpy(out_triplets, d_o, (size_t)m * 12, hipMemcpyDeviceToHost));
  *n_out = m;
  return DVS_OK;
}

dvs_status dvs_backproject(dvs_matcher* ctx, const dvs_keypoint* kps, int32_t n, const uint16_t* depth, int32_t rows, int32_t cols,
                           size_t step_bytes, float fx, float fy, float cx, float cy, const double* R, const double* t, double* world_xyz,
                           int32_t* out_index, int32_t* n_out) {
  DVS_ARG(ctx && n_out && n >= 0 && rows > 0 && cols > 0 && depth && R && t);
  *n_out = 0;
  if (n == 0) return DVS_OK;
  DVS_ARG(kps && world_xyz && out_index);
  // a keypoint whose rounded position lies outside the depth image is DROPPED, on this host entry point exactly as in
  // k_backproject / k_publish_keyframe / filterDepth's bounds check (the reference indexes the depth image unchecked there,
  // frontend.cpp:737: undefined behaviour, never a result to reproduce)
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  const size_t kb = ((size_t)n * sizeof(dvs_keypoint) + 15) & ~(size_t)15;
  uint8_t* base;
  DVS_TRY(matcher_scratch(ctx, 0, kb + 96 + (size_t)n * 24 + (size_t)n * 4 + 16 + (size_t)rows * cols * 2 + 32, (void**)&base));
  dvs_keypoint* d_k = (dvs_keypoint*)base;
  double* d_Rt = (double*)(base + kb);
  double* d_w = d_Rt + 12;
  int* d_oi = (int*)(d_w + 3 * (size_t)n); int* d_no = d_oi + n;
  uint16_t* d_dep = (uint16_t*)(((uintptr_t)(d_no + 2) + 15) & ~(uintptr_t)15);
  double Rt[12];
  memcpy(Rt, R, 72); memcpy(Rt + 9, t, 24);
  DVS_HIP(hipMemcpyAsync(d_k, kps, (size_t)n * sizeof(dvs_keypoint), hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpyAsync(d_Rt, Rt, 96, hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpy2DAsync(d_dep, (size_t)cols * 2, depth, step_bytes, (size_t)cols * 2, rows, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_backproject, dim3(1), dim3(256), 0, st, d_k, n, d_dep, rows, cols, (uint64_t)cols * 2, fx, fy, cx, cy, d_Rt, d_w, d_oi, d_no);
  DVS_HIP(hipGetLastError());
  int m = 0;
  DVS_HIP(hipMemcpyAsync(&m, d_no, 4, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipStreamSynchronize(st));
  if (m) {
    DVS_HIP(hipMemcpy(world_xyz, d_w, (size_t)m * 24, hipMemcpyDeviceToHost));
    DVS_HIP(hipMemcpy(out_index, d_oi, (size_t)m * 4, hipMemcpyDeviceToHost));
  }
  *n_out = m;
  return DVS_OK;
}

// snapshot association; optionally also hands back every observation's candidate list (landmarks with Hamming distance below the
// gate, in landmark order) so that a caller applying associations ONE BY ONE — the reference re-triangulates a landmark after each
// association (backend.cpp:758-777), which can move it before a later observation of the same keyframe is tested — re-evaluates
// exactly the observations whose candidates changed (include/dvslam/association.hpp)
static dvs_status associate_impl(dvs_matcher* ctx, const uint8_t* obs_desc, const float* obs_px, int32_t nobs, const uint8_t* lm_desc,
                                 const float* lm_xyz, int32_t nlm, const double* R, const double* t, double fx, double fy, double cx, double cy,
                                 double max_descriptor_distance, double max_reprojection_distance, int32_t* best, int64_t* cand_offsets,
                                 int32_t* cand_lm, int64_t cand_cap, int64_t* n_cand) {
  DVS_ARG(ctx && nobs >= 0 && nlm >= 0);
  if (n_cand) *n_cand = 0;
  if (cand_offsets) for (int i = 0; i <= nobs; i++) cand_offsets[i] = 0;
  if (nobs == 0) return DVS_OK;
  DVS_ARG(obs_desc && obs_px && best && R && t);
  for (int i = 0; i < nobs; i++) best[i] = -1;
  if (nlm == 0) return DVS_OK;
  DVS_ARG(lm_desc && lm_xyz);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  // (float)d < max_desc with integer d  <=>  d < ceil(max_desc)
  const int thr = (int)std::min<double>(ceil(max_descriptor_distance), 257.0);
  const long long* d_offs; const int* d_pairs; long long total = 0;
  DVS_TRY(matcher_thresh_device(ctx, obs_desc, nobs, lm_desc, nlm, thr, &d_offs, &d_pairs, &total));
  if (n_cand) *n_cand = total;
  if (total == 0) return DVS_OK;
  uint8_t* base;
  const size_t pb = ((size_t)nobs * 8 + 15) & ~(size_t)15, lb = ((size_t)nlm * 12 + 15) & ~(size_t)15;
  DVS_TRY(matcher_scratch(ctx, 2, pb + lb + 96 + (size_t)total * 8 + (size_t)nobs * 4 + 64, (void**)&base));
  float* d_px = (float*)base; float* d_lm = (float*)(base + pb);
  double* d_Rt = (double*)(base + pb + lb); double* d_err = d_Rt + 12;
  int* d_best = (int*)(d_err + total);
  double Rt[12];
  memcpy(Rt, R, 72); memcpy(Rt + 9, t, 24);
  DVS_HIP(hipMemcpyAsync(d_px, obs_px, (size_t)nobs * 8, hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpyAsync(d_lm, lm_xyz, (size_t)nlm * 12, hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpyAsync(d_Rt, Rt, 96, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_reproject_errors, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_pairs, total, d_px, d_lm, d_Rt, fx, fy, cx, cy, d_err);
  hipLaunchKernelGGL(k_assoc_argmin, dim3((nobs + 255) / 256), dim3(256), 0, st, d_offs, nobs, d_pairs, d_err, max_reprojection_distance, d_best);
  DVS_HIP(hipGetLastError());
  DVS_HIP(hipMemcpyAsync(best, d_best, (size_t)nobs * 4, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipStreamSynchronize(st));
  if (cand_offsets) {
    static_assert(sizeof(long long) == sizeof(int64_t), "offset width");
    DVS_HIP(hipMemcpy(cand_offsets, d_offs, ((size_t)nobs + 1) * 8, hipMemcpyDeviceToHost));
    if (cand_lm && total <= cand_cap) {
      std::vector<int> tri((size_t)total * 3);
      DVS_HIP(hipMemcpy(tri.data(), d_pairs, tri.size() * 4, hipMemcpyDeviceToHost));
      for (long long p = 0; p < total; p++) cand_lm[p] = tri[3 * (size_t)p + 1];
    } else if (cand_lm) {
      set_error("candidate list needs %lld entries, capacity %lld", total, (long long)cand_cap);
      return DVS_ERR_CAPACITY;
    }
  }
  return DVS_OK;
}

dvs_status dvs_associate(dvs_matcher* ctx, const uint8_t* obs_desc, const float* obs_px, int32_t nobs, const uint8_t* lm_desc,
                         const float* lm_xyz, int32_t nlm, const double* R, const double* t, double fx, double fy, double cx, double cy,
                         double max_descriptor_distance, double max_reprojection_distance, int32_t* best) {
  return associate_impl(ctx, obs_desc, obs_px, nobs, lm_desc, lm_xyz, nlm, R, t, fx, fy, cx, cy, max_descriptor_distance, max_reprojection_distance, best,
                        nullptr, nullptr, 0, nullptr);
}

dvs_status dvs_associate_candidates(dvs_matcher* ctx, const uint8_t* obs_desc, const float* obs_px, int32_t nobs, const uint8_t* lm_desc,
                                    const float* lm_xyz, int32_t nlm, const double* R, const double* t, double fx, double fy, double cx,
                                    double cy, double max_descriptor_distance, double max_reprojection_distance, int32_t* best,
                                    int64_t* cand_offsets, int32_t* cand_lm, int64_t cand_cap, int64_t* n_cand) {
  DVS_ARG(cand_offsets && n_cand);
  return associate_impl(ctx, obs_desc, obs_px, nobs, lm_desc, lm_xyz, nlm, R, t, fx, fy, cx, cy, max_descriptor_distance, max_reprojection_distance, best,
                        cand_offsets, cand_lm, cand_cap, n_cand);
}


// ---- Keyframe.msg (row N3)
static dvs_status kf_head(const dvs_keyframe_header* hdr, KfHead* H) {
  DVS_ARG(hdr && hdr->frame_id);
  const size_t sl = strlen(hdr->frame_id);
  if (sl + 1 > sizeof(H->frame_id)) { set_error("header.frame_id longer than %zu characters", sizeof(H->frame_id) - 1); return DVS_ERR_ARG; }
  memset(H, 0, sizeof(*H));
  H->sec = hdr->stamp_sec; H->nanosec = hdr->stamp_nanosec; H->slen = (uint32_t)sl + 1;
  memcpy(H->frame_id, hdr->frame_id, sl + 1);
  H->o1 = (12 + H->slen + 7u) & ~7u;
  H->keyframe_id = hdr->keyframe_id;
  for (int k = 0; k < 3; k++) H->pose[k] = hdr->translation[k];
  for (int k = 0; k < 4; k++) H->pose[3 + k] = hdr->rotation_xyzw[k];
  return DVS_OK;
}

size_t dvs_keyframe_cdr_capacity(const char* header_frame_id, int32_t n) {
  const size_t sl = header_frame_id ? strlen(header_frame_id) + 1 : 1;
  const size_t o2 = ((12 + sl + 7) & ~(size_t)7) + 64;
  return 4 + (n <= 0 ? o2 + 8 : o2 + 8 + 32 * (size_t)n + 8 + 64 * (size_t)(n - 1) + 60);
}

dvs_status dvs_publish_keyframe_device(dvs_matcher* ctx, const dvs_keyframe_header* hdr, const dvs_keypoint* d_kps, const uint8_t* d_desc,
                                       int32_t n, const uint16_t* d_depth, int32_t rows, int32_t cols, size_t step_bytes, float fx, float fy,
                                       float cx, float cy, const double* R, const double* t, uint8_t* d_out, size_t cap,
                                       uint64_t* d_out_size, int32_t* d_n_out) {
  DVS_ARG(ctx && n >= 0 && d_out && d_out_size && d_n_out && R && t && rows > 0 && cols > 0 && step_bytes >= (size_t)cols * 2);
  DVS_ARG(n == 0 || (d_kps && d_desc && d_depth));
  KfHead H;
  DVS_TRY(kf_head(hdr, &H));
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  double* d_Rt;
  DVS_TRY(matcher_scratch(ctx, 3, 96, (void**)&d_Rt));
  double Rt[12];
  memcpy(Rt, R, 72); memcpy(Rt + 9, t, 24);
  DVS_HIP(hipMemcpyAsync(d_Rt, Rt, 96, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_publish_keyframe, dim3(1), dim3(256), 0, st, H, d_kps, d_desc, n, d_depth, rows, cols, (uint64_t)step_bytes, fx, fy, cx, cy, d_Rt,
                     d_out, (unsigned long long)cap, (unsigned long long*)d_out_size, d_n_out);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status dvs_publish_keyframe(dvs_matcher* ctx, const dvs_keyframe_header* hdr, const dvs_keypoint* kps, const uint8_t* desc, int32_t n,
                                const uint16_t* depth, int32_t rows, int32_t cols, size_t step_bytes, float fx, float fy, float cx, float cy,
                                const double* R, const double* t, uint8_t* out, size_t cap, size_t* out_size, int32_t* n_landmarks) {
  DVS_ARG(ctx && hdr && out && out_size && n >= 0 && rows > 0 && cols > 0 && R && t);
  DVS_ARG(n == 0 || (kps && desc && depth));
  const size_t need = dvs_keyframe_cdr_capacity(hdr->frame_id, n);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  uint8_t* base;
  const size_t kb = ((size_t)n * sizeof(dvs_keypoint) + 63) & ~(size_t)63, db = ((size_t)n * 32 + 63) & ~(size_t)63,
               zb = ((size_t)rows * cols * 2 + 63) & ~(size_t)63, ob = (need + 63) & ~(size_t)63;
  DVS_TRY(matcher_scratch(ctx, 0, kb + db + zb + ob + 64, (void**)&base));
  dvs_keypoint* d_k = (dvs_keypoint*)base; uint8_t* d_d = base + kb; uint16_t* d_z = (uint16_t*)(base + kb + db);
  uint8_t* d_o = base + kb + db + zb; uint64_t* d_sz = (uint64_t*)(d_o + ob); int32_t* d_m = (int32_t*)(d_sz + 1);
  if (n) {
    DVS_HIP(hipMemcpyAsync(d_k, kps, (size_t)n * sizeof(dvs_keypoint), hipMemcpyHostToDevice, st));
    DVS_HIP(hipMemcpyAsync(d_d, desc, (size_t)n * 32, hipMemcpyHostToDevice, st));
    DVS_HIP(hipMemcpy2DAsync(d_z, (size_t)cols * 2, depth, step_bytes, (size_t)cols * 2, rows, hipMemcpyHostToDevice, st));
  }
  DVS_TRY(dvs_publish_keyframe_device(ctx, hdr, d_k, d_d, n, d_z, rows, cols, (size_t)cols * 2, fx, fy, cx, cy, R, t, d_o, need, d_sz, d_m));
  uint64_t sz = 0; int32_t m = 0;
  DVS_HIP(hipMemcpyAsync(&sz, d_sz, 8, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipMemcpyAsync(&m, d_m, 4, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipStreamSynchronize(st));
  *out_size = (size_t)sz;
  if (n_landmarks) *n_landmarks = m;
  if (sz > cap) { set_error("keyframe payload needs %llu bytes, buffer has %zu", (unsigned long long)sz, cap); return DVS_ERR_CAPACITY; }
  DVS_HIP(hipMemcpy(out, d_o, (size_t)sz, hipMemcpyDeviceToHost));
  return DVS_OK;
}

// the backend's side of the topic (keyframeCallback, backend.cpp): flat arrays ready for upload; host parsing, bounds checked
dvs_status dvs_keyframe_unpack_cdr(const uint8_t* buf, size_t len, dvs_keyframe_header* hdr, char* frame_id_buf, size_t frame_id_cap,
                                   uint64_t* landmark_ids, double* landmark_xyz, uint64_t* obs_landmark_ids, double* obs_pixels,
                                   uint8_t* obs_desc, int32_t cap_n, int32_t* n_landmarks, int32_t* n_observations) {
  DVS_ARG(buf && hdr && n_landmarks && n_observations && cap_n >= 0);
  *n_landmarks = *n_observations = 0;
  if (len < 4 || buf[0] != 0 || buf[1] != 1) { set_error("not a little-endian CDR payload"); return DVS_ERR_ARG; }
  const uint8_t* P = buf + 4;
  const size_t L = len - 4;
  size_t o = 0;
  auto need = [&](size_t nbytes) { return o + nbytes <= L; };
  auto align = [&](size_t a) { o = (o + a - 1) & ~(a - 1); };
  auto rd32 = [&](uint32_t* v) { align(4); if (!need(4)) return false; memcpy(v, P + o, 4); o += 4; return true; };
  auto rd64 = [&](void* v) { align(8); if (!need(8)) return false; memcpy(v, P + o, 8); o += 8; return true; };
  uint32_t sec, nsec, slen;
  bool ok = rd32(&sec) && rd32(&nsec) && rd32(&slen);
  if (!ok || slen == 0 || !need(slen) || P[o + slen - 1] != 0) { set_error("truncated or malformed header"); return DVS_ERR_ARG; }
  hdr->stamp_sec = (int32_t)sec; hdr->stamp_nanosec = nsec;
  if (frame_id_buf) {
    if (slen > frame_id_cap) { set_error("frame_id needs %u bytes", slen); return DVS_ERR_CAPACITY; }
    memcpy(frame_id_buf, P + o, slen);
    hdr->frame_id = frame_id_buf;
  } else hdr->frame_id = nullptr;
  o += slen;
  ok = rd64(&hdr->keyframe_id);
  for (int k = 0; k < 3 && ok; k++) ok = rd64(&hdr->translation[k]);
  for (int k = 0; k < 4 && ok; k++) ok = rd64(&hdr->rotation_xyzw[k]);
  uint32_t nl = 0, no = 0;
  ok = ok && rd32(&nl);
  if (!ok || nl > (L - o) / 32) { set_error("truncated landmark array"); return DVS_ERR_ARG; }
  if ((int64_t)nl > cap_n) { *n_landmarks = (int32_t)nl; set_error("%u landmarks, arrays hold %d", nl, cap_n); return DVS_ERR_CAPACITY; }
  for (uint32_t i = 0; i < nl && ok; i++) {
    uint64_t id; double x[3];
    ok = rd64(&id) && rd64(&x[0]) && rd64(&x[1]) && rd64(&x[2]);
    if (ok) { if (landmark_ids) landmark_ids[i] = id; if (landmark_xyz) memcpy(landmark_xyz + 3 * (size_t)i, x, 24); }
  }
  ok = ok && rd32(&no);
  if (!ok || no > (L - o) / 28) { set_error("truncated observation array"); return DVS_ERR_ARG; }
  if ((int64_t)no > cap_n) { *n_landmarks = (int32_t)nl; *n_observations = (int32_t)no; set_error("%u observations, arrays hold %d", no, cap_n); return DVS_ERR_CAPACITY; }
  for (uint32_t i = 0; i < no && ok; i++) {
    uint64_t id; double px, py; uint32_t dl;
    ok = rd64(&id) && rd64(&px) && rd64(&py) && rd32(&dl);
    if (!ok) break;
    if (dl != 32 || !need(32)) { set_error("observation %u carries a %u-byte descriptor (ORB rows are 32 bytes)", i, dl); return DVS_ERR_ARG; }
    if (obs_landmark_ids) obs_landmark_ids[i] = id;
    if (obs_pixels) { obs_pixels[2 * (size_t)i] = px; obs_pixels[2 * (size_t)i + 1] = py; }
    if (obs_desc) memcpy(obs_desc + 32 * (size_t)i, P + o, 32);
    o += 32;
  }
  if (!ok) { set_error("truncated payload"); return DVS_ERR_ARG; }
  *n_landmarks = (int32_t)nl; *n_observations = (int32_t)no;
  return DVS_OK;
}


// ---- Harris responses (row N4)
dvs_status dvs_harris_responses_device(dvs_matcher* ctx, const uint8_t* d_img, int32_t rows, int32_t cols, size_t step, const int32_t* d_x,
                                       const int32_t* d_y, int32_t n, int32_t block_size, float k, float* d_response) {
  DVS_ARG(ctx && n >= 0 && rows > 0 && cols > 0 && step >= (size_t)cols && block_size >= 1 && block_size <= 8);
  if (n == 0) return DVS_OK;
  DVS_ARG(d_img && d_x && d_y && d_response);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipLaunchKernelGGL(k_harris, dim3((n + 3) / 4), dim3(256), 0, matcher_stream(ctx), d_img, rows, cols, (uint64_t)step, d_x, d_y, n, block_size, k,
                     d_response);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status dvs_harris_responses(dvs_matcher* ctx, const uint8_t* img, int32_t rows, int32_t cols, size_t step, const int32_t* x,
                                const int32_t* y, int32_t n, int32_t block_size, float k, float* response) {
  DVS_ARG(ctx && n >= 0 && rows > 0 && cols > 0 && step >= (size_t)cols && block_size >= 1 && block_size <= 8);
  if (n == 0) return DVS_OK;
  DVS_ARG(img && x && y && response);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  uint8_t* base;
  const size_t ib = ((size_t)rows * cols + 63) & ~(size_t)63, nb = ((size_t)n * 4 + 63) & ~(size_t)63;
  DVS_TRY(matcher_scratch(ctx, 0, ib + 3 * nb, (void**)&base));
  int32_t* d_x = (int32_t*)(base + ib); int32_t* d_y = (int32_t*)(base + ib + nb); float* d_r = (float*)(base + ib + 2 * nb);
  DVS_HIP(hipMemcpy2DAsync(base, cols, img, step, cols, rows, hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpyAsync(d_x, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
  DVS_HIP(hipMemcpyAsync(d_y, y, (size_t)n * 4, hipMemcpyHostToDevice, st));
  DVS_TRY(dvs_harris_responses_device(ctx, base, rows, cols, (size_t)cols, d_x, d_y, n, block_size, k, d_r));
  DVS_HIP(hipMemcpyAsync(response, d_r, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  DVS_HIP(hipStreamSynchronize(st));
  return DVS_OK;
}

}  // extern "C"
