// orb_kernels.h — gfx950 kernels of the ORB extractor (included by orb.hip only).
//
// Kernels per batch of F frames (grid.z / grid.y = frame); orb.hip (enqueue_extract) holds the schedule:
//   k_resize4 / k_resize  x (nlevels-1)  pyramid, level l from level l-1 (k_pyr_cascade: all levels in one launch, <= 4 frames)
//                                                                                   ORBextractor.cpp:1169-1194
//   k_fast_wave<P> / k_fast_cell         FAST-9/16 score + per-cell NMS + threshold fallback, wavefront (workgroup) per cell
//                                                                                   ORBextractor.cpp:805-872
//   k_octree                             candidate gather + quad-tree distribution, workgroup per (frame, level)
//                                                                                   ORBextractor.cpp:555-779, 874-890
//   k_blur_stream / k_blur               7x7 fixed-point Gaussian of every level    ORBextractor.cpp:1132-1133
//   k_describe                           IC orientation + steered BRIEF + output    ORBextractor.cpp:76-146, 1142-1163
// The second name of a pair is the generic variant for inputs the fast one does not take (rows not dword aligned).
//
// Everything is integer or explicitly-rounded float32/float64 arithmetic; the file is compiled with
// -ffp-contract=off so no mul/add pair is fused behind our back.
#pragma once
#include <hip/hip_runtime.h>
#include "glibc_sincosf.h"
#include "lsort.h"
#include "orb_geom.h"
#include "orb_device_common.h"

namespace dvs {

typedef uint8_t u8;

struct ImgSrc {
  const u8* img0;     // level-0 frames (caller's buffer or our staging copy)
  uint64_t step0;     // bytes between rows of level 0
  uint64_t fstride0;  // bytes between frames of level 0
  u8* pyr;            // levels >= 1: pyr + f * frameBytes + lv[l].off
  // level-sharded extraction (small batches on several GPUs): only the levels in levelMask are processed, and the
  // descriptor stage writes keypoint i of level l to the fixed slot lv[l].kpOff + i of a level-slotted block instead of the
  // level-major compacted position (the other ranks' levels are gathered around it, dvs_orb_merge_levels_device)
  uint32_t levelMask; // bit l = level l is processed here (all ones: the whole frame)
  int32_t slotted;    // 1: slotted output + per-level counts, 0: the reference's compacted output + total
};

__device__ __forceinline__ const u8* level_ptr(const Geom* g, const ImgSrc& s, int f, int l, int& pitch) {
  if (l == 0) { pitch = (int)s.step0; return s.img0 + (uint64_t)f * s.fstride0; }
  pitch = g->lv[l].pitch;
  return s.pyr + (uint64_t)f * g->frameBytes + g->lv[l].off;
}

// latency-bound kernels of the chain behind FAST (quad-tree, descriptors, match): raised issue priority, so that a wave that has an
// instruction ready is not queued behind the throughput-bound waves it shares a SIMD with (experiment switch)
#ifndef DVS_CHAIN_PRIO_LEVEL
#define DVS_CHAIN_PRIO_LEVEL 0
#endif
#define DVS_CHAIN_PRIO() do { if (DVS_CHAIN_PRIO_LEVEL) __builtin_amdgcn_s_setprio(DVS_CHAIN_PRIO_LEVEL); } while (0)

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// XCD-aware work-item id (speed only): workgroups are dealt round-robin over the 8 XCDs, each with a private 4 MiB L2.  Remap
// the linear workgroup id so that every XCD works through ONE contiguous range of work items — for the (x = piece, y = frame)
// grids here that is a contiguous run of frames, so the lines neighbouring pieces share are fetched into one L2 once instead of
// into eight.  Bijective for any workgroup count.
__device__ __forceinline__ int xcd_contiguous_id() {
  const int nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// 24-bit multiplies, forced: hipcc lowers __umul24/__mul24 with a scalar or loop-invariant operand to v_mul_lo_u32 (+ v_add3),
// and the full 32-bit multiplier runs at a quarter of the 24-bit rate on gfx950.  Operands here are always < 2^24.
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// ... with a wave-uniform second factor taken straight from its SGPR (the "v" form costs a v_mov per use site)
__device__ __forceinline__ uint32_t mad_u24_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
  return d;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b) {  // (a[23:0] * b[23:0]) >> 32
  uint32_t d;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int mul_i24(int a, int b) {
  int d;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o);
    if (lane_id() >= o) v += t;
  }
  return v;
}

// exclusive scan over the 256 threads of a block; wsum = 5 ints of LDS.  Contains two barriers.
__device__ __forceinline__ int block_excl_scan256(int v, int* wsum, int& total) {
  int incl = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  __syncthreads();  // wsum may still be read from a previous call
  if (lane_id() == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { int s = wsum[i]; if (i < w) base += s; }
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + incl - v;
}

// the same for NT threads (NT / 64 wavefronts), wsum = NT / 64 + 1 ints
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int& total) {
  constexpr int NW = NT / 64;
  int incl = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  __syncthreads();  // wsum may still be read from a previous call
  if (lane_id() == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) { int s = wsum[i]; if (i < w) base += s; tot += s; }
  total = tot;
  return base + incl - v;
}

// threads per quad-tree workgroup.  The kernel is a chain of short block-wide passes over <= ~6000 candidates and <= ~450 nodes
// separated by barriers, i.e. latency-bound: 512 threads shorten it (0.117 -> 0.091 ms alone), but it runs beside the blur and
// the wave slots it then occupies cost the blur more than the quad-tree gains (81 k -> 79 k frames/s; 1024: 75 k)
constexpr int kOctT = 256;
constexpr int kOctTMax = 512;  // few frames: nothing competes for the wave slots, the tree alone decides the latency
// Threads a quad-tree workgroup actually uses.  The kernel's duration is its slowest workgroup — level 0 with ~6000 candidates —
// while the small levels finish early: in batch mode (more workgroups than CUs) the launch has 512 threads and the levels keep
// 512 / 256 / 128 of them (the other wavefronts exit at once and free their slots for the blur running beside), which
// shortens the long workgroups without taking more wave slots in total than 256 threads everywhere did.
__device__ __forceinline__ int oct_threads() {
  const int bd = (int)blockDim.x;
  if (gridDim.x * gridDim.y <= 256 || bd < 512) return bd;
  const int l = (int)blockIdx.y;   // grid = (frames, levels)
#ifdef DVS_OCT_GRADE_L0
  return l < 1 ? 512 : 256;
#else
  return l < 2 ? 512 : (l < 5 ? 256 : 128);
#endif
}
// exclusive scan over oct_threads() (multiple of 64, <= kOctTMax) threads; wsum = kOctTMax / 64 + 1 ints
__device__ __forceinline__ int block_excl_scan_rt(int v, int* wsum, int& total) {
  const int nw = oct_threads() >> 6;
  int incl = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  __syncthreads();  // wsum may still be read from a previous call
  if (lane_id() == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; i++) { int sv = wsum[i]; if (i < w) base += sv; tot += sv; }
  total = tot;
  return base + incl - v;
}

// The same with ONE barrier: the wavefront sums alternate between two buffers (`par`, uniform, flipped by every call), so a call never
// overwrites what a slower wavefront may still be reading from the call before — the readers of the buffer it writes passed the
// previous call's barrier.  wsum2 = 2 x (kOctTMax / 64 + 1) ints.  Unlike block_excl_scan_rt this is NOT a barrier in front of the caller's
// own earlier LDS writes being read by other threads — only behind them.
__device__ __forceinline__ int block_excl_scan_db(int v, int* wsum2, int& par, int& total) {
  const int nw = oct_threads() >> 6;
  const int incl = wave_incl_scan(v);
  const int w = threadIdx.x >> 6;
  int* ws = wsum2 + par * (kOctTMax / 64 + 1);
  par ^= 1;
  if (lane_id() == 63) ws[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; i++) { const int sv = ws[i]; if (i < w) base += sv; tot += sv; }
  total = tot;
  return base + incl - v;
}

// =============================================================================================
// pyramid: cv::resize(prev, cur, sz, 0, 0, INTER_LINEAR) on 8UC1 with OpenCV's 11-bit fixed-point
// coefficients (tables built on the host, orb.hip build_resize_tables).  4 output pixels / thread.
// =============================================================================================
__global__ __launch_bounds__(256) void k_resize(const u8* __restrict__ src, uint64_t sfs, int sw, int sh, int sp,
                                                u8* __restrict__ dst, uint64_t dfs, int dw, int dh, int dp,
                                                const int* __restrict__ xofs, const int* __restrict__ alpha,
                                                const int* __restrict__ yofs, const int* __restrict__ beta) {
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (x4 >= dw || y >= dh) return;
  const u8* s = src + (uint64_t)f * sfs;
  u8* d = dst + (uint64_t)f * dfs + (uint64_t)y * dp;
  const int sy = yofs[y];
  const int b = beta[y];
  const int b0 = (int)(short)(b & 0xffff), b1 = b >> 16;
  const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
  const u8* p0 = s + (uint64_t)r0 * sp;
  const u8* p1 = s + (uint64_t)r1 * sp;
  uint32_t out = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = x4 + i;
    if (x < dw) {
      const int sx = xofs[x];
      const int sx1 = min(sx + 1, sw - 1);
      const int a = alpha[x];
      const int a0 = (int)(short)(a & 0xffff), a1 = a >> 16;
      const int h0 = p0[sx] * a0 + p0[sx1] * a1;
      const int h1 = p1[sx] * a0 + p1[sx1] * a1;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xff) << (8 * i);
    }
  }
  if (x4 + 3 < dw) {
    *reinterpret_cast<uint32_t*>(d + x4) = out;
  } else {
    for (int i = 0; i < 4 && x4 + i < dw; i++) d[x4 + i] = (u8)(out >> (8 * i));
  }
}

// Same arithmetic, 4 output pixels per thread from TWO 12-byte source windows (dword loads) instead of 16 byte
// gathers: v_alignbit moves the group's first tap to byte 0 of an 8-byte window (all 8 taps of the group lie inside it for
// scale factors < 1.5), then per pixel ONE v_perm picks (left, right) into 16-bit halves and ONE v_dot2_u32_u16 applies the
// Q11 pair.  Whenever the reference clamps the right tap (last column) its coefficient is 0, so its byte is irrelevant.
constexpr int kResizeRows = 8;  // output rows per thread: one table entry, 2 x kResizeRows independent 12-byte windows in flight, 7 of the 16 row windows shared (4 rows: 3 of 8; step -0.6 % in same-box alternating runs)
__global__ __launch_bounds__(256) void k_resize4(const u8* __restrict__ src, uint64_t sfs, int sw, int sh, int sp,
                                                 u8* __restrict__ dst, uint64_t dfs, int dw, int dh, int dp,
                                                 const ResizeGroup* __restrict__ xt, const int* __restrict__ yofs,
                                                 const int* __restrict__ beta, int guardFrame, int tilesX, uint32_t magicTiles, uint32_t magicX,
                                                 int ngx, int fpg, uint32_t magicNgx, int nimg) {
  // (raising these waves' issue priority over the FAST waves they run beside, s_setprio, was measured: 3 % slower overall)
  // grid = (tiles of a frame, frames), workgroups dealt to the XCDs frame by frame (xcd_contiguous_id): the source rows and the
  // 128-byte lines neighbouring tiles share then meet in ONE L2 (round 2's (x, y, frame) grid spread a frame's tiles over all eight:
  // 1.32x the algorithmic bytes fetched).  The two divisions are by host-made reciprocals on the scalar unit (exact: the host checks
  // the ranges and passes magic = 0 otherwise).
  // The x axis of the grid runs over the 4-pixel groups of `fpg` consecutive frames, one row of groups behind the other (ngx = dw / 4
  // groups per frame): level widths are no multiples of a wavefront's 256 pixels — 130 groups at level 5 filled 3 wavefronts to two
  // thirds, one lane in six of the whole chain idled — while 8 x 130 groups fill 17 wavefronts.  Rows stay wave-uniform; the frame is a
  // per-lane offset.  fpg = frames / 8 for batches of whole multiples of 8 (a group is what one XCD receives), else 1.
  const uint32_t wg = (uint32_t)xcd_contiguous_id(), tiles = gridDim.x;
  const uint32_t grp = magicTiles ? __umulhi(wg, magicTiles) : wg / tiles;
  const uint32_t tile = wg - grp * tiles;
  const uint32_t by = magicX ? __umulhi(tile, magicX) : tile / (uint32_t)tilesX;
  const uint32_t bx = tile - by * (uint32_t)tilesX;
  const uint32_t q = bx * 64u + threadIdx.x;
  const uint32_t fl = magicNgx ? __umulhi(q, magicNgx) : q / (uint32_t)ngx;   // frame within the group
  const int gx = (int)(q - fl * (uint32_t)ngx);
  const int x4 = gx * 4;
  const uint32_t f = grp * (uint32_t)fpg + fl;
  // a wavefront is one threadIdx.y row of the (64, 4) block: everything that depends on y only is wave-uniform, and saying so
  // (v_readfirstlane) moves the row table lookups and the 64-bit row address arithmetic to the scalar unit
  const int y0 = ((int)by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * kResizeRows;
  if (fl >= (uint32_t)fpg || f >= (uint32_t)nimg || y0 >= dh) return;
  const ResizeGroup t = xt[gx];
  // uniform group bases + 32-bit per-lane offsets (the frames of a group span far less than 4 GB: checked on the host): no 64-bit
  // vector multiply-adds.  The lane's frame rides in its column offsets.
  const u8* s = src + (uint64_t)(grp * (uint32_t)fpg) * sfs;
  u8* d = dst + (uint64_t)(grp * (uint32_t)fpg) * dfs;
  const uint32_t fos = fl * (uint32_t)sfs, fod = fl * (uint32_t)dfs + (uint32_t)x4;
  // the 12-byte windows of a row's last group run up to 11 bytes past the row: harmless everywhere (the bytes past the last tap carry
  // zero weights, and what follows a row is the next row, the next level or the allocation's slack) except behind the LAST row of
  // the LAST frame of a caller's buffer (guardFrame).  Round 2a sent every row-end lane — and with it its whole wavefront, one in
  // five at level 1, one in two at level 7 — through the bytewise path.
  const bool tailLane = t.base + 12 > sw;
  // Consecutive output rows share source rows (scale 1.2: the upper source row of output row y + 1 is the lower one of row y five times
  // out of six): a wavefront's rows are uniform, so whether row r's upper source row is row r - 1's lower one is a scalar condition —
  // then it is neither loaded nor filtered horizontally again (3 of the 8 row windows and horizontal passes of a thread, typically)
  uint32_t w0[kResizeRows][3], w1[kResizeRows][3];
  int bb[kResizeRows];
  bool shared[kResizeRows];
  int prevLower = -1;
#pragma unroll
  for (int r = 0; r < kResizeRows; r++) {
    const int y = min(y0 + r, dh - 1);
    const int sy = yofs[y];
    bb[r] = beta[y];
    const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
    shared[r] = r0 == prevLower;
    prevLower = r1;
    const u8* p0 = s + ((uint32_t)(r0 * sp) + ((uint32_t)t.base + fos));
    const u8* p1 = s + ((uint32_t)(r1 * sp) + ((uint32_t)t.base + fos));
    const bool fastw = !(tailLane && (int)f == guardFrame && r1 == sh - 1);
    if (fastw) {
      const uint2 c = *reinterpret_cast<const uint2*>(p1);
      w1[r][0] = c.x; w1[r][1] = c.y; w1[r][2] = *reinterpret_cast<const uint32_t*>(p1 + 8);
      if (!shared[r]) {
        const uint2 a = *reinterpret_cast<const uint2*>(p0);
        w0[r][0] = a.x; w0[r][1] = a.y; w0[r][2] = *reinterpret_cast<const uint32_t*>(p0 + 8);
      } else {
        w0[r][0] = w0[r][1] = w0[r][2] = 0u;
      }
    } else {  // row tail: bytewise, never past the last valid pixel
      int lastpx = sw - 1 - t.base;
      asm volatile("" : "+v"(lastpx));   // defined HERE: the compiler otherwise hoists the twelve clamps below into the path every wavefront takes
#pragma unroll
      for (int k = 0; k < 3; k++) {
        uint32_t u = 0, v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int o = min(4 * k + i, lastpx);
          u |= (uint32_t)p0[o] << (8 * i);
          v |= (uint32_t)p1[o] << (8 * i);
        }
        w0[r][k] = u; w1[r][k] = v;
      }
    }
  }
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  uint32_t hlow[4] = {0u, 0u, 0u, 0u};   // the horizontal sums (h >> 4 << 4: what the vertical term uses) of the previous output row's lower source row
#pragma unroll
  for (int r = 0; r < kResizeRows; r++) {
    if (y0 + r >= dh) break;
    // the row's two Q11 weights (0 .. 2048 for INTER_LINEAR), pre-shifted on the scalar unit: with B = b << 12 (< 2^24) and
    // H = h & ~15 (h >> 4 << 4, < 2^20), v_mul_hi_u32_u24(B, H) = (b * (h >> 4) * 2^16) >> 32 = (b * (h >> 4)) >> 16 — OpenCV's
    // vertical term in two instructions (and, mul_hi) instead of three (shift, mul, shift)
    const uint32_t b0 = (uint32_t)(bb[r] & 0xffff) << 12, b1 = (uint32_t)(bb[r] >> 16) << 12;
    // 8-byte window starting at the group's first tap
    uint32_t hup[4];
    if (shared[r]) {
#pragma unroll
      for (int i = 0; i < 4; i++) hup[i] = hlow[i];
    } else {
      const uint32_t lo0 = __builtin_amdgcn_alignbit(w0[r][1], w0[r][0], t.shift), hi0 = __builtin_amdgcn_alignbit(w0[r][2], w0[r][1], t.shift);
#pragma unroll
      for (int i = 0; i < 4; i++)   // (left tap, right tap) as two u16 halves, then a0 * left + a1 * right in one v_dot2_u32_u16
        hup[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi0, lo0, t.sel[i])), __builtin_bit_cast(us2, t.alpha[i]), 0u, false) & ~15u;
    }
    const uint32_t lo1 = __builtin_amdgcn_alignbit(w1[r][1], w1[r][0], t.shift), hi1 = __builtin_amdgcn_alignbit(w1[r][2], w1[r][1], t.shift);
    uint32_t sum[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t h0 = hup[i];   // already without its low four bits (both branches above)
      const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi1, lo1, t.sel[i])), __builtin_bit_cast(us2, t.alpha[i]), 0u, false) & ~15u;
      hlow[i] = h1;
      sum[i] = mulhi_u24(b0, h0) + mulhi_u24(b1, h1) + 2u;   // the pixel is sum >> 2 (<= 255: a convex combination of bytes), sum < 1024
    }
    // four sums -> four bytes in five instructions: the sums of a pixel pair side by side in 16-bit halves, ONE shift for both (the
    // neighbour's two low bits land in byte 1, which nobody reads), one permute picks bytes 0 and 2 of either pair
    const uint32_t p01 = (sum[0] | (sum[1] << 16)) >> 2, p23 = (sum[2] | (sum[3] << 16)) >> 2;
    const uint32_t out = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
    // dw is the padded width (multiple of 4, <= pitch): mirrored columns included
    *reinterpret_cast<uint32_t*>(d + ((uint32_t)((y0 + r) * dp) + fod)) = out;
  }
}

// ---------------------------------------------------------------------------------------------
// Pyramid cascade: ALL levels in one launch.  The 7 dependent per-level launches are short and latency-bound (each
// waits on table -> source -> store round trips, ~15 us even for the 72 k-pixel last level); here a workgroup stages
// one level-0 region in LDS and walks down the levels with LDS as source, writing only the pixels it owns.  Same
// integer arithmetic as k_resize (OpenCV's Q11 INTER_LINEAR), level 0 is read ~1.1x, every level is written once.
// ---------------------------------------------------------------------------------------------
constexpr int kPyrTabX = 1024, kPyrTabY = 640;  // LDS table entries over all levels of one tile (checked on the host)
template <int NT>
__global__ __launch_bounds__(NT) void k_pyr_cascade(const Geom* __restrict__ g, const PyrTile* __restrict__ tiles, ImgSrc src,
                                                     const int* __restrict__ xofs, const int* __restrict__ alpha,
                                                     const int* __restrict__ yofs, const int* __restrict__ beta, int bufBytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  // coefficient tables of every level's region, made tile-relative once (one global-latency phase for the whole cascade)
  __shared__ uint2 tx[kPyrTabX];  // .x = left tap | right tap << 16 (columns relative to the source region), .y = a0 | a1 << 16
  __shared__ uint2 ty[kPyrTabY];  // .x = top row | bottom row << 16 (relative),                              .y = b0 | b1 << 16
  const PyrTile& T = tiles[blockIdx.x];
  const int f = blockIdx.y;
  const int tid = threadIdx.x;
  const int nl = g->nlevels;
  u8* buf[2] = {psm, psm + bufBytes};
  // stage the level-0 region (sx0 is a multiple of 4; rows are dword aligned — checked on the host)
  const int sw = T.sx1 - T.sx0, shh = T.sy1 - T.sy0;
  int sP = (sw + 3) & ~3;
  {
    const u8* img = src.img0 + (uint64_t)f * src.fstride0 + (uint64_t)T.sy0 * src.step0 + T.sx0;
    const int wpr = sP >> 2;
    uint32_t* d32 = reinterpret_cast<uint32_t*>(buf[0]);
    const int n = wpr * shh;
    const float inv = 1.0f / (float)wpr;
    const int W0 = g->lv[0].w;
    // four words per thread and trip, all requested before the first is stored (a trip per word paid the global latency a dozen times)
    auto fetch = [&](int i) -> uint32_t {
      const int r = (int)(((float)i + 0.5f) * inv);
      const int c = i - r * wpr;
      const u8* p = img + (uint64_t)r * src.step0 + 4 * c;
      if (T.sx0 + 4 * c + 3 < W0) return *reinterpret_cast<const uint32_t*>(p);
      uint32_t v = 0;   // last word of a row whose width is not a multiple of 4: never read past the image
      for (int b = 0; b < 4; b++) if (T.sx0 + 4 * c + b < W0) v |= (uint32_t)p[b] << (8 * b);
      return v;
    };
    for (int i0 = tid; i0 < n; i0 += 4 * NT) {
      uint32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = i0 + NT * u < n ? fetch(i0 + NT * u) : 0u;
#pragma unroll
      for (int u = 0; u < 4; u++) if (i0 + NT * u < n) d32[i0 + NT * u] = v[u];
    }
    // tables: every level's entries are REQUESTED before any is stored (one global round trip for the whole cascade — level by level the
    // two dependent trips per level, tile record -> table entry, were most of the kernel's 35 us on one frame).  A thread holds one x and
    // one y entry per level: regions are at most 256 wide and high (checked on the host); up to 15 levels below level 0.
    constexpr int kLv = DVS_MAX_LEVELS - 1;
    int xv[kLv], av[kLv], yv[kLv], bv[kLv], xoK[kLv], yoK[kLv], oxK[kLv], oyK[kLv], swK[kLv], shK[kLv];
    bool hx[kLv], hy[kLv];
    int xo = 0, yo = 0, ox = T.sx0, oy = T.sy0;
    bool live = true;
#pragma unroll
    for (int k = 1; k <= kLv; k++) {
      hx[k - 1] = hy[k - 1] = false;
      if (k < nl && live) {
        const PyrTileLevel R = T.lv[k];
        const int cw = R.cx1 - R.cx0, chh = R.cy1 - R.cy0;
        if (cw <= 0 || chh <= 0) live = false;
        else {
          const LevelGeom& L = g->lv[k];
          const int cwp = (cw + 3) & ~3;
          xoK[k - 1] = xo; yoK[k - 1] = yo; oxK[k - 1] = ox; oyK[k - 1] = oy; swK[k - 1] = g->lv[k - 1].w - 1; shK[k - 1] = g->lv[k - 1].h - 1;
          if (tid < cwp) { const int x = min(R.cx0 + tid, L.w - 1); xv[k - 1] = xofs[L.xtab + x]; av[k - 1] = alpha[L.xtab + x]; hx[k - 1] = true; }
          if (tid < chh) { yv[k - 1] = yofs[L.ytab + R.cy0 + tid]; bv[k - 1] = beta[L.ytab + R.cy0 + tid]; hy[k - 1] = true; }
          xo += cwp; yo += chh; ox = R.cx0; oy = R.cy0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kLv; k++) {
      if (hx[k]) { const int o = xv[k]; tx[xoK[k] + tid] = make_uint2((uint32_t)(o - oxK[k]) | ((uint32_t)(min(o + 1, swK[k]) - oxK[k]) << 16), (uint32_t)av[k]); }
      if (hy[k]) { const int sy = yv[k]; ty[yoK[k] + tid] = make_uint2((uint32_t)(min(max(sy, 0), shK[k]) - oyK[k]) | ((uint32_t)(min(max(sy + 1, 0), shK[k]) - oyK[k]) << 16), (uint32_t)bv[k]); }
    }
  }
  __syncthreads();
  int cur = 0, xo = 0, yo = 0;
  for (int k = 1; k < nl; k++) {
    const PyrTileLevel R = T.lv[k];
    const int cw = R.cx1 - R.cx0, chh = R.cy1 - R.cy0;
    if (cw <= 0 || chh <= 0) break;  // deeper levels own nothing either
    const LevelGeom& L = g->lv[k];
    const int dP = (cw + 3) & ~3;
    const u8* sb = buf[cur];
    u8* db = buf[cur ^ 1];
    const int ng = dP >> 2;
    const int rpp = NT / ng;  // rows per pass
    const int gx = tid % ng, ry = tid / ng;
    u8* gdst = src.pyr + (uint64_t)f * g->frameBytes + L.off;
    if (ry < rpp) {
      int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint2 e = tx[xo + 4 * gx + i];
        sx[i] = (int)(e.x & 0xffff); sx1[i] = (int)(e.x >> 16);
        a0[i] = (int)(short)(e.y & 0xffff); a1[i] = (int)e.y >> 16;
      }
      const int x0 = R.cx0 + 4 * gx;
      const bool full = x0 >= R.ox0 && x0 + 3 < R.ox1;
      const bool mirror = x0 + 3 >= L.w - 9;
      for (int y = R.cy0 + ry; y < R.cy1; y += rpp) {
        const uint2 e = ty[yo + y - R.cy0];
        const int b0 = (int)(short)(e.y & 0xffff), b1 = (int)e.y >> 16;
        const u8* p0 = sb + (int)(e.x & 0xffff) * sP;
        const u8* p1 = sb + (int)(e.x >> 16) * sP;
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int h0 = p0[sx[i]] * a0[i] + p0[sx1[i]] * a1[i];
          const int h1 = p1[sx[i]] * a0[i] + p1[sx1[i]] * a1[i];
          const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
          out |= (uint32_t)(v & 0xff) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(db + (y - R.cy0) * dP + 4 * gx) = out;
        if (y >= R.oy0 && y < R.oy1) {
          u8* grow = gdst + (uint64_t)y * L.pitch;
          if (full) {
            *reinterpret_cast<uint32_t*>(grow + x0) = out;
          } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (x0 + i >= R.ox0 && x0 + i < R.ox1) grow[x0 + i] = (u8)(out >> (8 * i));
          }
          // BORDER_REFLECT_101 continuation right of the image (read by the streaming blur): column 2W-2-c mirrors c
          if (mirror) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const int c = x0 + i;
              if (c >= R.ox0 && c < R.ox1 && c >= L.w - 9 && c <= L.w - 2) grow[2 * L.w - 2 - c] = (u8)(out >> (8 * i));
            }
          }
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    sP = dP; xo += dP; yo += chh;
  }
}

// =============================================================================================
// FAST-9/16 per reference cell.  One 256-thread workgroup = one cell of one frame:
//   1. stage the (cw x ch) sub-image into LDS
//   2. cheap opposite-pair rejection test at minTh for every interior pixel, survivors compacted into
//      an LDS work list with a wave ballot (lanes stay dense for the expensive part)
//   3. exact corner score S = max over 9-arcs (cornerScore<16>); a pixel is a FAST corner at threshold
//      t  <=>  S >= t, so ONE score map serves both thresholds (20 and 7)
//   4. 3x3 non-max suppression inside the cell interior (outside = 0, as cv::FAST on the sub-image),
//      iniTh keypoints if any survive, otherwise minTh keypoints (ORBextractor.cpp:843-846),
//      emitted row-major with a block scan so candidate order equals the reference's.
// =============================================================================================
__device__ __forceinline__ int fast_corner_score(const u8* c, int v) {
  // ring offsets of cv::FAST pattern 16 (fast_score.cpp makeOffsets), k = 0..15
  constexpr int P = kTilePitch;
  const int o[16] = {3 * P,      3 * P + 1,  2 * P + 2,  P + 3,  3,  -P + 3,  -2 * P + 2,  -3 * P + 1,
                     -3 * P,     -3 * P - 1, -2 * P - 2, -P - 3, -3, P - 3,   2 * P - 2,   3 * P - 1};
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
    A = max(A, min(min(lo3[k], lo3[(k + 3) & 15]), lo3[(k + 6) & 15]));  // min of d[k..k+8]
    B = min(B, max(max(hi3[k], hi3[(k + 3) & 15]), hi3[(k + 6) & 15]));  // max of d[k..k+8]
  }
  return max(A, -B) - 1;
}

__global__ __launch_bounds__(256) void k_fast_cell(const Geom* __restrict__ g, const Cell* __restrict__ cells, ImgSrc src,
                                                   uint32_t* __restrict__ cand, int* __restrict__ cellCount, int cell0) {
  constexpr int P = kTilePitch;
  __shared__ __attribute__((aligned(16))) u8 tile[kMaxCellDim * P];
  __shared__ __attribute__((aligned(16))) u8 score[kMaxCellDim * P];
  __shared__ uint16_t work[(kMaxCellDim - 6) * (kMaxCellDim - 6)];
  __shared__ int nwork;
  __shared__ int wsum[5];
  __shared__ int any20;

  const int tid = threadIdx.x;
  const int f = blockIdx.y;
  const int ci = cell0 + blockIdx.x;
  const Cell cell = cells[ci];
  const LevelGeom& L = g->lv[cell.level];
  int pitch;
  const u8* img = level_ptr(g, src, f, cell.level, pitch);
  const int cw = cell.cw, ch = cell.ch;
  const int iw = cw - 6, ih = ch - 6;
  int* countOut = cellCount + (uint64_t)f * g->totalCells + ci;
  if (iw <= 0 || ih <= 0) {  // cv::FAST finds nothing in a sub-image narrower than 7
    if (tid == 0) *countOut = 0;
    return;
  }
  if (tid == 0) { nwork = 0; any20 = 0; }
  // 1. stage
  {
    const u8* base = img + (uint64_t)cell.y0 * pitch + cell.x0;
    const int n = cw * ch;
    const float inv = 1.0f / (float)cw;
    for (int p = tid; p < n; p += 256) {
      int y = (int)(((float)p + 0.5f) * inv);
      int x = p - y * cw;
      tile[y * P + x] = base[(uint64_t)y * pitch + x];
    }
    uint32_t* s32 = reinterpret_cast<uint32_t*>(score);
    for (int p = tid; p < ch * P / 4; p += 256) s32[p] = 0;
  }
  __syncthreads();
  // one score map serves both of the reference's calls when minTh <= iniTh; with minTh > iniTh the second call finds a subset of the
  // first's (empty) result, so the lower of the two thresholds is the one to test and score at
  const int tini = g->iniTh, tmin = min(g->minTh, tini);
  const int npx = iw * ih;
  const float invw = 1.0f / (float)iw;
  // 2. rejection test + compaction
  for (int p0 = 0; p0 < npx; p0 += 256) {
    const int p = p0 + tid;
    bool pass = false;
    int c = 0;
    if (p < npx) {
      int y = (int)(((float)p + 0.5f) * invw);
      int x = p - y * iw;
      c = (y + 3) * P + (x + 3);
      const int v = tile[c];
      const int lo = v - tmin, hi = v + tmin;
      auto cls = [&](int off) -> int { int r = tile[c + off]; return (r < lo ? 1 : 0) | (r > hi ? 2 : 0); };
      int dbits = cls(3 * P) | cls(-3 * P);             // ring 0 / 8
      dbits &= cls(3) | cls(-3);                        // ring 4 / 12
      dbits &= cls(2 * P + 2) | cls(-2 * P - 2);        // ring 2 / 10
      dbits &= cls(-2 * P + 2) | cls(2 * P - 2);        // ring 6 / 14
      pass = dbits != 0;
    }
    const unsigned long long m = __ballot(pass);
    if (m) {
      int base = 0;
      const int leader = __ffsll((long long)m) - 1;
      if (lane_id() == leader) base = atomicAdd(&nwork, __popcll(m));
      base = __shfl(base, leader);
      if (pass) work[base + __popcll(m & ((1ull << lane_id()) - 1ull))] = (uint16_t)c;
    }
  }
  __syncthreads();
  // 3. exact score for the survivors
  const int nw = nwork;
  for (int e = tid; e < nw; e += 256) {
    const int c = work[e];
    const int s = fast_corner_score(&tile[c], (int)tile[c]);
    if (s >= tmin) score[c] = (u8)s;
  }
  __syncthreads();
  // 4. NMS + ordered emission.  Thread t owns the contiguous row-major pixel range [t*k, (t+1)*k).
  const int k = (npx + 255) >> 8;  // <= 19
  uint32_t m20 = 0, m7 = 0;
  const int pbeg = tid * k;
  for (int i = 0; i < k; i++) {
    const int p = pbeg + i;
    if (p >= npx) break;
    int y = (int)(((float)p + 0.5f) * invw);
    int x = p - y * iw;
    const int c = (y + 3) * P + (x + 3);
    const int s = score[c];
    if (s > 0) {
      // pixels outside the interior were never written: they hold 0, exactly cv::FAST's zeroed buffers
      const bool ismax = s > score[c - 1] && s > score[c + 1] && s > score[c - P - 1] && s > score[c - P] &&
                         s > score[c - P + 1] && s > score[c + P - 1] && s > score[c + P] && s > score[c + P + 1];
      if (ismax) { m7 |= 1u << i; if (s >= tini) m20 |= 1u << i; }
    }
  }
  if (m20) any20 = 1;  // benign race: every writer stores 1
  __syncthreads();
  const uint32_t sel = any20 ? m20 : m7;
  int total;
  int rank = block_excl_scan256(__popc(sel), wsum, total);
  uint32_t* out = cand + (uint64_t)f * g->candPerFrame + L.candOff + (uint64_t)cell.slot * L.cellCap;
  const int cap = L.cellCap;
  for (int i = 0; i < k; i++) {
    if (sel & (1u << i)) {
      const int p = pbeg + i;
      int y = (int)(((float)p + 0.5f) * invw);
      int x = p - y * iw;
      const int c = (y + 3) * P + (x + 3);
      // sub-image coords (x+3, y+3) shifted by j*wCell, i*hCell (ORBextractor.cpp:865-866)
      if (rank < cap) out[rank] = pack_pt(x + 3 + cell.j * L.wCell, y + 3 + cell.i * L.hCell, score[c]);
      rank++;
    }
  }
  if (tid == 0) *countOut = min(total, cap);
}

// ---------------------------------------------------------------------------------------------
// Wave-per-cell variant (used whenever level rows are dword aligned).  Same results as k_fast_cell,
// restructured for the instruction-issue bound the first version hit:
//   * one WAVEFRONT per cell, four independent cells per workgroup -> no workgroup barriers at all;
//     LDS operations of one wave execute in issue order, phases are separated by wave-level fences
//   * the cell is staged with aligned dword loads and kept at the same byte phase in LDS
//   * rejection test on 4 pixels per lane: 11 aligned LDS dword reads + v_alignbyte windows replace
//     36 byte reads; "every opposite pair has a darker (brighter) sample" is evaluated as
//     max_pairs(min(d_k, d_k+8)) < -t  (min_pairs(max(..)) > t): 4 ops per pair, no per-sample classes
//   * survivors (~17 % of pixels) are compacted IN ORDER by ballots, scored exactly, and the corners
//     (~3 %) compacted again; NMS and the threshold fallback then touch only that short list.
// ---------------------------------------------------------------------------------------------
// start-up probe: does the LDS-DMA take global addresses that are not dword aligned on this device / driver configuration?
// (it does on gfx950 under ROCm's unaligned-access mode; if it ever masked the low address bits the tiles would be wrong, so
// k_fast_wave only relies on it when this returned the exact bytes)
__global__ void k_probe_lds_dma(const u8* __restrict__ src, uint32_t* __restrict__ out, int shift) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64];
  const int lane = (int)threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + 4 * lane),
                                   (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[lane] = lds[lane];
}

// ---------------------------------------------------------------------------------------------
// What keeps the instruction count down (hipcc -S listings; round 1's register-staged, one-survivor-per-lane form of the same
// phases retired 36 % more wave-level VALU instructions per cell):
//   * staging by LDS-DMA: the tile's LDS image is dword-linear in lane order, so each global_load_lds_dword wave-instruction
//     lands 64 consecutive dwords with no VGPR round trip (no ds_write, no per-row 64-bit address arithmetic); the score tile
//     is cleared with 16-byte stores
//   * score stage on TWO survivors per lane: the two pixels' ring samples are packed into the halves of one register and the
//     3+3+3 window network runs on v_pk_minimum3_f16 / v_pk_maximum3_f16 (gfx950) — byte values 0..255 are f16 denormals whose
//     order is their integer order, min / max never round — so one instruction serves two pixels
//   * compaction: one packed 4-bit mask per lane and a DPP prefix sum instead of four ballots and eight v_mbcnt
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// exact cornerScore<16> of two pixels at once (pixel a in the low halves, b in the high halves).  ta / tb point 3 rows and 3
// columns BEFORE the pixel, so that every ring offset is a non-negative ds_read immediate (no per-sample address arithmetic).
template <int P>
__device__ __forceinline__ i16x2 fast_corner_score2(const u8* ta, const u8* tb) {
  constexpr int C = 3 * P + 3;
  const int o[16] = {C + 3 * P,  C + 3 * P + 1,  C + 2 * P + 2,  C + P + 3,  C + 3,  C - P + 3,  C - 2 * P + 2,  C - 3 * P + 1,
                     C - 3 * P,  C - 3 * P - 1,  C - 2 * P - 2,  C - P - 3,  C - 3,  C + P - 3,  C + 2 * P - 2,  C + 3 * P - 1};
  f16x2 r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    u16x2 t;
    t.x = ta[o[k]]; t.y = tb[o[k]];
    r[k] = __builtin_bit_cast(f16x2, t);
  }
  u16x2 vv;
  vv.x = ta[C]; vv.y = tb[C];
  auto mn3 = [](f16x2 a, f16x2 b, f16x2 c) -> f16x2 { return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c); };
  auto mx3 = [](f16x2 a, f16x2 b, f16x2 c) -> f16x2 { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); };
  f16x2 lo3[16], hi3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lo3[k] = mn3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
    hi3[k] = mx3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
  }
  f16x2 lo9[16], hi9[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lo9[k] = mn3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    hi9[k] = mx3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
  }
  f16x2 mn[6], mx[6];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    mn[k] = mn3(hi9[3 * k], hi9[3 * k + 1], hi9[3 * k + 2]);
    mx[k] = mx3(lo9[3 * k], lo9[3 * k + 1], lo9[3 * k + 2]);
  }
  mn[5] = hi9[15]; mx[5] = lo9[15];
  const f16x2 minHi = mn3(mn3(mn[0], mn[1], mn[2]), mn3(mn[3], mn[4], mn[5]), mn[5]);
  const f16x2 maxLo = mx3(mx3(mx[0], mx[1], mx[2]), mx3(mx[3], mx[4], mx[5]), mx[5]);
  const i16x2 v = __builtin_bit_cast(i16x2, vv);
  const i16x2 a = v - __builtin_bit_cast(i16x2, minHi), b = __builtin_bit_cast(i16x2, maxLo) - v;
  const i16x2 one = {1, 1};
  return __builtin_elementwise_max(a, b) - one;
}

// inclusive prefix sum over the wavefront by DPP row shifts / broadcasts (six v_add_u32_dpp)
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1 (zero fill: one v_add_u32_dpp, no separate move)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

template <int P>
__global__ __launch_bounds__(256) void k_fast_wave(const Geom* __restrict__ g, const Cell* __restrict__ cells, ImgSrc src,
                                                   uint32_t* __restrict__ cand, int* __restrict__ cellCount, int cell0, int cell1,
                                                   uint32_t magicGX) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const int lane = lane_id();
  const int wvi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index as an SGPR: the cell, its level geometry and
  // frames -> XCDs contiguously (xcd_contiguous_id): the workgroups of one frame — whose cells share halo rows and 128-byte lines —
  // then meet in ONE L2 instead of eight (HBM fetch 2.1x -> see profiles of the algorithmic bytes; FAST is not fetch-bound, the
  // time is unchanged)
  const uint32_t xid = (uint32_t)xcd_contiguous_id();
  // frame = xid / gridDim.x by the host's reciprocal (exact for this grid; 0 = not representable): the division proper is ~25 scalar and
  // 4 vector instructions at the head of every wavefront, before its first load can leave
  const int f = (int)(magicGX ? __umulhi(xid, magicGX) : xid / gridDim.x);
  const int ci = cell0 + (int)(xid - (uint32_t)f * gridDim.x) * 4 + wvi;       // every size derived from them stay on the scalar unit
  if (ci >= cell1) return;                                           // cells [cell0, cell1) of the level-major cell table
  unsigned char* base = fsm + wvi * g->fastWaveLds;
  u8* tile = base;
  constexpr int T = fast_tile_bytes(P);   // tile and score tile: a compile-time distance apart, so a pixel's score is an immediate offset
  u8* score = base + T;                   // from its tile address
  uint16_t* work = reinterpret_cast<uint16_t*>(base + T + g->fastRows * P);   // the score tile is only as tall as the tallest cell
  const Cell cell = cells[ci];
  const LevelGeom& L = g->lv[cell.level];
  int pitch;
  const u8* img = level_ptr(g, src, f, cell.level, pitch);
  const int cw = cell.cw, ch = cell.ch;
  const int iw = cw - 6, ih = ch - 6;
  int* countOut = cellCount + (uint64_t)f * g->totalCells + ci;
  if (iw <= 0 || ih <= 0) { if (lane == 0) *countOut = 0; return; }
  // tile origin: the dword holding the cell's first column — or, where the DMA engine takes byte addresses, one column left of it,
  // which puts the first interior column (x0 + 3) on a dword of the tile for EVERY cell: 31 interior columns then are 8 four-pixel
  // groups instead of 8 or 9 by the cell's phase, and a 32-row cell is 4 trips of the rejection loop instead of 4.5 on average
  const int xa = g->fastByteDma ? cell.x0 - 1 : (cell.x0 & ~3), ox = cell.x0 - xa;
  // 1. stage by LDS-DMA.  One wave-instruction lands RP whole tile rows (RP * W consecutive LDS dwords, W = P / 4 dwords per
  //    row; lanes >= RP * W idle): lane -> (row-in-piece, dword column) is fixed, so a lane's global offset is computed once
  //    and each further piece only advances the SCALAR base by RP rows — no vector arithmetic in the loop beyond the
  //    row-bound compare of the last piece.  Columns past the cell's last dword re-read that dword (never outside the image).
  {
    constexpr int W = P / 4, RP = 64 / W;
    const int wpr = (ox + cw + 3) >> 2;  // dwords per row the cell needs (<= W)
    const int lr = lane / W, lc = lane - lr * W;
    const uint32_t off = (uint32_t)mad_i24(lr, pitch, min(lc, wpr - 1) * 4);
    const u8* rb = img + (uint64_t)cell.y0 * pitch + xa;
    const uint32_t tileLds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) u8*)tile;
    // the piece's global address as SCALAR base + the lane's 32-bit offset, its LDS base in M0: no vector instruction per piece (the
    // builtin's form added the 64-bit base to a register pair and compared the row per piece)
    auto piece = [&](int r0) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                   :: "s"(tileLds0 + (uint32_t)(r0 * P)), "v"(off), "s"(rb + (uint64_t)r0 * pitch) : "memory", "m0");
    };
    if (lane < RP * W) {
      int r0 = 0;
      for (; r0 + RP <= ch; r0 += RP) piece(r0);
      if (r0 + lr < ch) piece(r0);   // the last rows
    }
    uint4* s128 = reinterpret_cast<uint4*>(score);
    for (int i = lane; i < (ch * P) >> 4; i += 64) s128[i] = make_uint4(0u, 0u, 0u, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  wave_lds_fence();
  const int tmin = g->minTh, tini = g->iniTh;
  // 2. rejection test, 4 pixels per lane.  Lane = (row of the trip, dword column): the column group, its border mask and the
  //    lane's offsets are fixed for the whole cell, a trip advances every lane by the same number of rows, so the loop carries no
  //    index arithmetic (round 2a derived row and column from a flat index each trip: ~11 of its 122 instructions)
  const int cx0 = ox + 3, cx1 = ox + cw - 3;  // interior columns in tile coordinates
  const int g0 = cx0 >> 2, ng = ((cx1 - 1) >> 2) - g0 + 1;
  const int rpt = cell.rpt & 0xFF;            // rows per trip = 64 / ng (ng <= 19 for cells up to 76 pixels), host-made with 1 / ng
  const int tailRows = cell.rpt >> 8;         // ... and the rows of the last trip
  const int rsub = (int)(((float)lane + 0.5f) * cell.inv_ng), gi = lane - rsub * ng;
  // column mask of this lane's group: bit j = pixel j of the group is an interior column (first / last group only partly)
  // — as the sign bit of one byte per pixel, the form the loop's test results arrive in
  uint32_t cmLane = rsub < rpt ? 0x80808080u : 0u;
  if (gi == 0) cmLane &= 0x80808080u << (8 * (cx0 & 3));
  if (gi == ng - 1) cmLane &= 0x80808080u >> (8 * (3 - ((cx1 - 1) & 3)));
  const uint32_t cmTail = rsub < tailRows ? cmLane : 0u;   // the last trip's mask: only the rows that exist
  const uint32_t* const rpFirst = reinterpret_cast<const uint32_t*>(tile) + mad_i24(rsub + 3, P / 4, g0 + gi);   // centre word of the lane's first row
  // survivors are listed by the offset of the pixel 3 rows and 3 columns up-left; the lane's four candidates entries ride in the
  // halves of two registers (ds_write_b16 / ds_write_b16_d16_hi store either half), advanced by one packed add each per trip
  const int cbase0 = mad_i24(rsub, P, (g0 + gi) * 4 - 3);
  const u16x2 c01First = {(uint16_t)cbase0, (uint16_t)(cbase0 + 1)}, c23First = {(uint16_t)(cbase0 + 2), (uint16_t)(cbase0 + 3)};
  const uint16_t cstep = (uint16_t)(rpt * P);
  const u16x2 cstep2 = {cstep, cstep};
  const uint32_t workLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)work;
  const uint32_t* work32 = reinterpret_cast<const uint32_t*>(work);
  const uint32_t tileLds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) u8*)tile;
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  // The reference's two literal calls (ORBextractor.cpp:826-846): cv::FAST at iniThFAST, and ONLY for a cell that ends with no corner
  // again at minThFAST.  Every phase below runs at the pass's threshold tcur: survivors of the rejection test, exact scores, the corner
  // list (score >= tcur) and the strict 3x3 maximum among exactly those corners — the score tile holds nothing else: pass 0 leaves the
  // scores of corners >= iniTh there, which the second pass, whose corners are a superset, writes again with the same values.
  // (Round 1-4 ran ONE pass at minTh and picked the >= iniTh maxima afterwards: the same corners for iniTh >= minTh, but every cell
  // scored its minTh survivors — 2.5 trips of the score stage per cell against 1.5 at iniTh on the benchmark's frames, where 6 % of the
  // cells are empty at iniTh and pay a second rejection loop.)
  int ncorner = 0;
  int tcur = tini;
  for (int pass = 0;; pass++) {
  int nwork = 0;
  const uint32_t* rp = rpFirst;
  u16x2 c01 = c01First, c23 = c23First;
  const s16x2 T2 = {(short)tcur, (short)tcur};
  for (int row0 = 0; row0 < ih; row0 += rpt, rp += rpt * (P / 4), c01 += cstep2, c23 += cstep2) {
    const uint32_t Om3 = rp[-3 * (P / 4)], Op3 = rp[3 * (P / 4)];
    const uint32_t Lm2 = rp[-2 * (P / 4) - 1], Om2 = rp[-2 * (P / 4)], Rm2 = rp[-2 * (P / 4) + 1];
    const uint32_t L0 = rp[-1], O0 = rp[0], R0 = rp[1];
    const uint32_t Lp2 = rp[2 * (P / 4) - 1], Op2 = rp[2 * (P / 4)], Rp2 = rp[2 * (P / 4) + 1];
    // packed 16-bit evaluation, two pixels per instruction: with r_k the raw ring samples,
    //   all four opposite pairs hold a darker sample   <=>  max_pairs(min(r_k, r_k+8)) < v - t
    //   all four opposite pairs hold a brighter sample <=>  min_pairs(max(r_k, r_k+8)) > v + t
    // A ring sample of the pixel pair (2 hh, 2 hh + 1) is two bytes of the eight bytes (hi : lo) of two neighbouring words: ONE byte
    // permute moves them into the 16-bit halves (round 2a shifted the words first — six v_alignbyte per trip — and unpacked then)
    uint32_t sgn[2];
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      auto pick = [&](uint32_t hi, uint32_t lo, int shift) -> s16x2 {   // bytes shift + 2 hh, shift + 2 hh + 1 of (hi : lo)
        const uint32_t sel = (uint32_t)(shift + 2 * hh) | 0x0c00u | ((uint32_t)(shift + 2 * hh + 1) << 16) | 0x0c000000u;
        return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(hi, lo, sel));
      };
      const s16x2 v2 = pick(0u, O0, 0);
      const s16x2 r0 = pick(0u, Op3, 0), r8 = pick(0u, Om3, 0);
      const s16x2 r4 = pick(R0, O0, 3), r12 = pick(O0, L0, 1);             // ring 4 ( 3, 0), ring 12 (-3, 0)
      const s16x2 r2 = pick(Rp2, Op2, 2), r14 = pick(Op2, Lp2, 2);         // ring 2 ( 2, 2), ring 14 (-2, 2)
      const s16x2 r6 = pick(Rm2, Om2, 2), r10 = pick(Om2, Lm2, 2);         // ring 6 ( 2,-2), ring 10 (-2,-2)
      // the four pair minima / maxima and their fold on the packed f16 pipe (byte values are f16 denormals: their order is the integer
      // order and min / max never round — as in the score stage): the 3-operand forms fold four values in two instructions
      auto f = [](s16x2 a) { return __builtin_bit_cast(f16x2, a); };
      auto mn2 = [&](s16x2 a, s16x2 b) { return __builtin_elementwise_minimum(f(a), f(b)); };
      auto mx2 = [&](s16x2 a, s16x2 b) { return __builtin_elementwise_maximum(f(a), f(b)); };
      const f16x2 mnf = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_elementwise_maximum(mn2(r0, r8), mn2(r4, r12)), mn2(r2, r10)), mn2(r6, r14));
      const f16x2 mxf = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_elementwise_minimum(mx2(r0, r8), mx2(r4, r12)), mx2(r2, r10)), mx2(r6, r14));
      const s16x2 mn = __builtin_bit_cast(s16x2, mnf), mx = __builtin_bit_cast(s16x2, mxf);
      // mn < v - t or mx > v + t  <=>  max(v - mn, mx - v) > t: the sign of t - max(...)
      sgn[hh] = __builtin_bit_cast(uint32_t, T2 - __builtin_elementwise_max(v2 - mn, mx - v2));
    }
    // the four sign bytes (bytes 1 and 3 of either half) side by side: bit 8 j + 7 = pixel j passes; gated by the lane's column
    // mask (the cell's last trip: cut down to the rows that exist)
    const uint32_t m4 = __builtin_amdgcn_perm(sgn[1], sgn[0], 0x07050301u) & (row0 + rpt >= ih ? cmTail : cmLane);
#ifndef DVS_EXP_NO_EMPTY_TRIP_SKIP
    // a trip without a survivor (23 % of them at iniTh = 20 on the benchmark's frames: flat regions under sensor noise) skips the
    // count, the prefix sum and the four conditional stores
    if (__builtin_amdgcn_ballot_w64(m4 != 0u) == 0ull) continue;
#endif
    const int cnt = __popc(m4);
    const int incl = wave_incl_scan_dpp(cnt);
    // the four conditional stores, by hand: per pixel ONE compare (SDWA picks the pixel's byte) and ONE add — the running address,
    // which starts one slot before the lane's first and is advanced under the store's own execution mask.  (The compiler's form of
    // the same: mask + compare + address add + copy + entry add per pixel.)  LDS operations of a wave complete in order, so the
    // loads the compiler schedules around this block are unaffected.
    uint32_t waS = workLds - 2u + 2u * (uint32_t)nwork;   // scalar part (kept apart: two vector instructions for the address, not three)
    asm volatile("" : "+s"(waS));
    uint32_t wa = waS + 2u * (uint32_t)(incl - cnt);
    unsigned long long sv;
    asm volatile(
        "v_cmp_ne_u32_sdwa vcc, %[m], %[z] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_add_u32_e32 %[a], 2, %[a]\n\t"
        "ds_write_b16 %[a], %[c01]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmp_ne_u32_sdwa vcc, %[m], %[z] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_add_u32_e32 %[a], 2, %[a]\n\t"
        "ds_write_b16_d16_hi %[a], %[c01]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmp_ne_u32_sdwa vcc, %[m], %[z] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_add_u32_e32 %[a], 2, %[a]\n\t"
        "ds_write_b16 %[a], %[c23]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmp_ne_u32_sdwa vcc, %[m], %[z] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_add_u32_e32 %[a], 2, %[a]\n\t"
        "ds_write_b16_d16_hi %[a], %[c23]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [a] "+v"(wa), [sv] "=&s"(sv)
        : [m] "v"(m4), [z] "s"(0), [c01] "v"(c01), [c23] "v"(c23)
        : "vcc", "memory");
    nwork += __builtin_amdgcn_readlane(incl, 63);
  }
  wave_lds_fence();
  // 3. exact score for the survivors, two per lane (2 lane, 2 lane + 1); corners (score >= tcur) are re-compacted in place, row-major
  ncorner = 0;
  for (int e0 = 0; e0 < nwork; e0 += 128) {
    const int ea = e0 + 2 * lane;
    // idle halves read ONE harmless pixel (offset 0: a broadcast; stale entries would scatter over the banks)
    const uint32_t cc = ea < nwork ? work32[(e0 >> 1) + lane] : 0u;
    const uint32_t ca = cc & 0xFFFFu, cb = ea < nwork - 1 ? cc >> 16 : 0u;
    const i16x2 sc2 = fast_corner_score2<P>(tile + ca, tile + cb);
    // corners: list entry in range and score >= tcur.  The compare builtins ARE the ballots (a ballot of a combined predicate is
    // re-made by two more vector instructions), combined on the scalar unit
    const unsigned long long ba = __builtin_amdgcn_sicmp(ea, nwork, 40 /* < */) & __builtin_amdgcn_sicmp((int)sc2.x, tcur, 39 /* >= */);
    const unsigned long long bb = __builtin_amdgcn_sicmp(ea, nwork - 1, 40) & __builtin_amdgcn_sicmp((int)sc2.y, tcur, 39);
    // corners before this lane's: four v_mbcnt; the list keeps the survivors' form of the offset (3 rows and 3 columns up-left)
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ba >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ba,
                          __builtin_amdgcn_mbcnt_hi((uint32_t)(bb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb, 0u))));
    uint32_t la = 2u * rank + (workLds + 2u * (uint32_t)ncorner);
    const uint32_t pa = tileLds + ca, pb = tileLds + cb;
    unsigned long long sv;
    // both stores of either pixel under ITS ballot as the execution mask: score byte (the pixel's own byte of the score tile: an
    // immediate offset from its tile address; pixel b's score is the register's high half), list entry at the running address
    asm volatile(
        "s_and_saveexec_b64 %[sv], %[ba]\n\t"
        "ds_write_b8 %[pa], %[sc] offset:%[K]\n\t"
        "ds_write_b16 %[la], %[ca]\n\t"
        "v_add_u32_e32 %[la], 2, %[la]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_and_saveexec_b64 %[sv], %[bb]\n\t"
        "ds_write_b8_d16_hi %[pb], %[sc] offset:%[K]\n\t"
        "ds_write_b16 %[la], %[cb]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [la] "+v"(la), [sv] "=&s"(sv)
        : [ba] "s"(ba), [bb] "s"(bb), [pa] "v"(pa), [pb] "v"(pb), [sc] "v"(__builtin_bit_cast(uint32_t, sc2)), [ca] "v"(ca), [cb] "v"(cb),
          [K] "n"(T + 3 * P + 3)
        : "memory");
    ncorner += __popcll(ba) + __popcll(bb);
  }
  wave_lds_fence();
  // 4. NMS on the corner list: bit 14 = strict 3x3 maximum among the pass's corners
  bool anyMax = false;
  for (int e0 = 0; e0 < ncorner; e0 += 64) {
    const int e = e0 + lane;
    bool ismax = false;
    if (e < ncorner) {
      const int c = work[e];
      const u8* sp = score + c + (3 * P + 3);   // the corner's own byte
      const int sc = sp[0];
      // all eight neighbours read at once (a short-circuit chain is eight dependent LDS round trips), folded by four 3-operand maxima
      const int n0 = sp[-1], n1 = sp[1], n2 = sp[-P - 1], n3 = sp[-P], n4 = sp[-P + 1], n5 = sp[P - 1], n6 = sp[P], n7 = sp[P + 1];
      ismax = sc > max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
      work[e] = (uint16_t)(c | (ismax ? 0x4000 : 0));
    }
    anyMax = anyMax || (__ballot(ismax) != 0ull);
  }
  wave_lds_fence();
  // a second call at a threshold that is not lower finds a subset of nothing
  if (anyMax || pass == 1 || tmin >= tini) break;
  tcur = tmin;
  }
  // 5. ordered emission
  uint32_t* out = cand + (uint64_t)f * g->candPerFrame + L.candOff + (uint64_t)cell.slot * L.cellCap;
  const int cap = L.cellCap;
  const int selbit = 0x4000;
  int nout = 0;
  for (int e0 = 0; e0 < ncorner; e0 += 64) {
    const int e = e0 + lane;
    const int w = e < ncorner ? work[e] : 0;
    const bool sel = (w & selbit) != 0;
    const unsigned long long b = __ballot(sel);
    if (sel) {
      const int c = w & 0x3fff;                    // offset of the pixel 3 rows and 3 columns up-left of the corner
      const int y3 = c / P, xx = c - y3 * P + 3 - ox, yy = y3 + 3;  // sub-image coordinates (the column never carries: x - 3 + 3 < P)
      const int rank = nout + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
      if (rank < cap) out[rank] = pack_pt(xx + cell.j * L.wCell, yy + cell.i * L.hCell, score[c + (3 * P + 3)]);
    }
    nout += __popcll(b);
  }
  if (lane == 0) *countOut = min(nout, cap);
}

// =============================================================================================
// quad-tree distribution (DistributeOctTree).  One workgroup per (level, frame).  The reference's
// std::list is kept as an ARRAY IN LIST ORDER that is rebuilt by prefix sums after every sweep:
//   full sweep : every multi-point node is split; new list = reverse(children in creation order) ++
//                surviving single-point nodes in old order           (push_front + erase, :622-681)
//   ordered    : nodes sorted by (count, UL.x) with the libstdc++ introsort replica (ties!), split
//                from the back until size >= N                        (:689-753)
// Points never move: each keeps the list position of its node (nodeOf); a node's winner is the max
// response with the lowest candidate index (:757-776), resolved with one atomicMax per point.
// =============================================================================================
#define OCT_T (oct_threads())  /* threads of a quad-tree workgroup that take part (see oct_threads) */
#ifdef DVS_QT_PROF   /* tools/qt_phase_profile.sh: 100 MHz time stamps of the level-0 tree of frame 0, phase by phase */
__device__ unsigned long long g_qt_prof[96];
#define QT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (i) < 96) g_qt_prof[i] = wall_clock64(); } while (0)
#else
#define QT_STAMP(i) do {} while (0)
#endif
struct QNode { int16_t ulx, uly, brx, bry; int32_t cnt; int32_t pt; };

__device__ __forceinline__ int qt_quadrant(const QNode& nd, int x, int y) {
  const int mx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)   (:482)
  const int my = nd.uly + ((nd.bry - nd.uly + 1) >> 1);
  return (x < mx ? 0 : 1) | (y < my ? 0 : 2);  // 0:n1 1:n2 2:n3 3:n4  (:514-524)
}
__device__ __forceinline__ QNode qt_child(const QNode& nd, int q, int cnt) {
  const int mx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1);
  const int my = nd.uly + ((nd.bry - nd.uly + 1) >> 1);
  QNode c;
  c.ulx = (int16_t)((q & 1) ? mx : nd.ulx);
  c.brx = (int16_t)((q & 1) ? nd.brx : mx);
  c.uly = (int16_t)((q & 2) ? my : nd.uly);
  c.bry = (int16_t)((q & 2) ? nd.bry : my);
  c.cnt = cnt;
  c.pt = -1;
  return c;
}

struct QtShared {
  QNode *n0, *n1;  // the two node arrays; selected by value (a runtime-indexed array member would put the struct in scratch)
  __device__ __forceinline__ QNode* nodes_(int c) const { return c ? n1 : n0; }
  int* childCnt;   // 4 per node
  int* posArr;     // split node: first list position of its children; unsplit: -(newpos+1)
  int* flag;       // per node: 1 = split in this pass
  int* expl;       // multi-point nodes in creation order (list positions)
  int* ecum;       // phase B: inclusive sum of non-empty children in processing order
  unsigned long long* sortbuf;
};

// What a point needs of its node to find its quadrant: the split point, both coordinates in one word.  The point loops of a sweep
// are bound by the workgroup's LDS traffic (level 0: ~4000 points x 4 passes per sweep), so they read this word — and, in
// qt_rebuild, one 8-byte record — instead of the 16-byte node and its four child counters (44 -> 16 bytes per point).
__device__ __forceinline__ uint32_t qt_mid(const QNode& nd) {
  const int mx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1), my = nd.uly + ((nd.bry - nd.uly + 1) >> 1);   // as qt_quadrant / qt_child
  return (uint32_t)mx | ((uint32_t)my << 16);
}
__device__ __forceinline__ int qt_quadrant_mid(uint32_t mid, int x, int y) {
  return (x < (int)(mid & 0xFFFFu) ? 0 : 1) | (y < (int)(mid >> 16) ? 0 : 2);
}

// counts children of every multi-point node of the current list; `mids` = S words of scratch (posArr: dead between two rebuilds)
__device__ __forceinline__ void qt_count_children(const QNode* nodes, int S, int* childCnt, uint32_t* mids, const uint32_t* pts,
                                                  const int* nodeOf, int n) {
  for (int k = threadIdx.x; k < 4 * S; k += OCT_T) childCnt[k] = 0;
  for (int k = threadIdx.x; k < S; k += OCT_T) mids[k] = qt_mid(nodes[k]);
  __syncthreads();
  // four points per thread and trip: the dependent LDS chain (node-of-point -> node -> counter) of one point is ~400 cycles of latency
  // and a thread of the level-0 workgroup walks ~24 points; with the loads of four points issued together a trip costs one chain
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * OCT_T) {
    int k[4];
    uint32_t p[4], md[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = i0 + u * OCT_T; k[u] = i < n ? nodeOf[i] : -1; p[u] = i < n ? pts[i] : 0u; }
#pragma unroll
    for (int u = 0; u < 4; u++) md[u] = mids[max(k[u], 0)];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k[u] >= 0) atomicAdd(&childCnt[4 * k[u] + qt_quadrant_mid(md[u], pt_x(p[u]), pt_y(p[u]))], 1);
  }
  __syncthreads();
}

__device__ __forceinline__ int qt_mask(const int* childCnt, int k) {
  return (childCnt[4 * k] > 0 ? 1 : 0) | (childCnt[4 * k + 1] > 0 ? 2 : 0) | (childCnt[4 * k + 2] > 0 ? 4 : 0) |
         (childCnt[4 * k + 3] > 0 ? 8 : 0);
}

// After flag[]/posArr[] of split nodes are set (posArr = first child position) and T = number of new
// children: place unsplit nodes behind the children in old order, write the new node array, re-point
// the points.  Returns the number of unsplit nodes (the new list holds T + that many).
// flag[k] / posArr[k] of a split node must be visible to the thread that owns k (k = tid, tid + OCT_T, ...) on entry.
__device__ __forceinline__ int qt_rebuild(QtShared& sh, int cur, int S, int T, uint32_t* pts, int* nodeOf, int n, int* wsum2, int& par) {
  const QNode* old = sh.nodes_(cur);
  QNode* nw = sh.nodes_(cur ^ 1);
  // unsplit nodes: stable compaction behind the children block
  int carry = 0;
  for (int b = 0; b < S; b += OCT_T) {
    const int k = b + threadIdx.x;
    const int u = (k < S && !sh.flag[k]) ? 1 : 0;
    int tot;
    const int ex = block_excl_scan_db(u, wsum2, par, tot);
    if (u) sh.posArr[k] = -((T + carry + ex) + 1);
    carry += tot;
  }
  // (no barrier: a thread reads flag / posArr of its own nodes only — the scan above and the caller's loops map k to threads alike)
  // per node: its successors in the new list, and the 8-byte record its points read below (the sort buffer is dead here):
  //   low word = split point; high word = first child's position | non-empty children << 16 | single-point children << 20 | 1 << 24
  //   (split node), or the node's own new position (unsplit)
  unsigned long long* rec = sh.sortbuf;
  for (int k = threadIdx.x; k < S; k += OCT_T) {
    const QNode nd = old[k];
    if (sh.flag[k]) {
      const int base = sh.posArr[k];
      const int c0 = sh.childCnt[4 * k], c1 = sh.childCnt[4 * k + 1], c2 = sh.childCnt[4 * k + 2], c3 = sh.childCnt[4 * k + 3];
      const int mask = (c0 > 0 ? 1 : 0) | (c1 > 0 ? 2 : 0) | (c2 > 0 ? 4 : 0) | (c3 > 0 ? 8 : 0);
      const int single = (c0 == 1 ? 1 : 0) | (c1 == 1 ? 2 : 0) | (c2 == 1 ? 4 : 0) | (c3 == 1 ? 8 : 0);
      const int cc[4] = {c0, c1, c2, c3};
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (mask & (1 << q)) nw[base + __popc(mask >> (q + 1))] = qt_child(nd, q, cc[q]);
      rec[k] = (unsigned long long)qt_mid(nd) | ((unsigned long long)((uint32_t)base | (uint32_t)mask << 16 | (uint32_t)single << 20 | 1u << 24) << 32);
    } else {
      const int np = -(sh.posArr[k] + 1);
      nw[np] = nd;
      rec[k] = (unsigned long long)(uint32_t)np << 32;
    }
  }
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * OCT_T) {   // four points per trip, loads first (see qt_count_children)
    int k[4];
    uint32_t p[4];
    unsigned long long rc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = i0 + u * OCT_T; k[u] = i < n ? nodeOf[i] : -1; p[u] = i < n ? pts[i] : 0u; }
#pragma unroll
    for (int u = 0; u < 4; u++) rc[u] = rec[max(k[u], 0)];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * OCT_T;
      if (k[u] < 0) continue;
      const uint32_t hi = (uint32_t)(rc[u] >> 32);
      if (!(hi & (1u << 24))) { nodeOf[i] = (int)hi; continue; }
      const int q = qt_quadrant_mid((uint32_t)rc[u], pt_x(p[u]), pt_y(p[u]));
      const int np = (int)(hi & 0xFFFFu) + __popc(((hi >> 16) & 15u) >> (q + 1));
      if ((hi >> (20 + q)) & 1u) { nw[np].pt = i; nodeOf[i] = -1; }
      else nodeOf[i] = np;
    }
  }
  __syncthreads();
  return carry;
}

// creation-ordered list of multi-point nodes among the T freshly created children (positions T-1..0)
__device__ __forceinline__ int qt_build_expand_list(QtShared& sh, int cur, int T, int* wsum) {
  const QNode* nodes = sh.nodes_(cur);
  int carry = 0;
  for (int b = 0; b < T; b += OCT_T) {
    const int j = b + threadIdx.x;
    const int pos = T - 1 - j;
    const int fl = (j < T && nodes[pos].cnt > 1) ? 1 : 0;
    int tot;
    const int ex = block_excl_scan_rt(fl, wsum, tot);
    if (fl) sh.expl[carry + ex] = pos;
    carry += tot;
  }
  __syncthreads();
  return carry;
}

// ---- workgroup std::sort (lsort.h, "restated so that it parallelises"): ranges are partitioned one wavefront each, round by
// round (the ranges of a round are disjoint), leaves of <= 16 elements are placed by rank, one lane per element.
// a: m packed 64-bit elements in LDS, key = a >> 12.  Lp / Rp: scratch of m ints each (LDS).  Result in place: the sorted
// elements' LOW 32 BITS (payload + the low key bits); the upper key bits are dropped.
constexpr int kSortRanges = 128;  // > kMaxQuota / 17 live ranges of more than 16 elements
struct SortShared { uint32_t rng[2][kSortRanges]; int cnt[2]; };

__device__ __forceinline__ void qt_sort_block(unsigned long long* a, int m, int* Lp, int* Rp, SortShared& ss) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // leaf descriptor per element, kept in Lp (a finished range's scratch is never used again): first | end << 16
  auto emit = [&](int f, int l, int d, int nxt) {
    if (l - f > 16) {
      if (lane == 0) ss.rng[nxt][atomicAdd(&ss.cnt[nxt], 1)] = (uint32_t)f | ((uint32_t)l << 12) | ((uint32_t)d << 24);
    } else if (lane < l - f) {
      Lp[f + lane] = f | (l << 16);
    }
  };
  if (tid == 0) { ss.cnt[0] = 0; ss.cnt[1] = 0; }
  __syncthreads();
  if (wv == 0) {
    int lg = 0;
    for (int q = m; q > 1; q >>= 1) lg++;
    emit(0, m, 2 * lg, 0);
  }
  __syncthreads();
  for (int cur = 0;; cur ^= 1) {
    const int cnt = ss.cnt[cur];
    if (cnt == 0) break;
    for (int ri = wv; ri < cnt; ri += OCT_T / 64) {
      const uint32_t r = ss.rng[cur][ri];
      const int f = (int)(r & 0xFFFu), l = (int)((r >> 12) & 0xFFFu), d = (int)(r >> 24);
      if (d == 0) {  // depth limit: heapsort the range (std::__partial_sort), every element its own leaf
        if (lane == 0) lsort::heap_sort_all(a + f, a + l, lsort::Less<12>());
        for (int i = f + lane; i < l; i += 64) Lp[i] = i | ((i + 1) << 16);
        continue;
      }
      const int cut = wave_partition<12>(a, Lp, Rp, f, l, lane);
      emit(f, cut, d - 1, cur ^ 1);
      emit(cut, l, d - 1, cur ^ 1);
    }
    __syncthreads();
    if (tid == 0) ss.cnt[cur] = 0;
    __syncthreads();
  }
  // leaves: stable placement by rank.  Only the payload (low 32 bits: the list index the callers read back) is carried to the
  // sorted position, through Rp / Lp, so that no element has to be held in registers across the barrier
  for (int i = tid; i < m; i += OCT_T) {
    const int fl = Lp[i];
    const int f = fl & 0xFFFF, l = fl >> 16;
    const unsigned long long ke = a[i] >> 12;
    int rnk = f;
    for (int j = f; j < l; j++) {
      const unsigned long long kj = a[j] >> 12;
      rnk += (kj < ke || (kj == ke && j < i)) ? 1 : 0;
    }
    Rp[i] = rnk;
  }
  __syncthreads();
  for (int i = tid; i < m; i += OCT_T) Lp[Rp[i]] = (int)(unsigned)a[i];
  __syncthreads();
  for (int i = tid; i < m; i += OCT_T) a[i] = (unsigned long long)(unsigned)Lp[i];
  __syncthreads();
}

#ifdef DVS_TEST_HOOKS
__global__ __launch_bounds__(kOctTMax) void k_test_sort(unsigned long long* __restrict__ v, int m) {
  __shared__ unsigned long long a[kMaxQuota];
  __shared__ int Lp[kMaxQuota], Rp[kMaxQuota];
  __shared__ SortShared ss;
  for (int i = threadIdx.x; i < m; i += OCT_T) a[i] = v[i];
  __syncthreads();
  qt_sort_block(a, m, Lp, Rp, ss);
  for (int i = threadIdx.x; i < m; i += OCT_T) v[i] = a[i];
}
#endif

__device__ __forceinline__ void octree_body(const Geom* __restrict__ g, const uint32_t* __restrict__ cand,
                                            const int* __restrict__ cellCount, int* __restrict__ cellOff,
                                            uint32_t* __restrict__ ptsAll, int* __restrict__ nodeOfAll,
                                            int* __restrict__ candTotal, uint32_t* __restrict__ lvlKp,
                                            int* __restrict__ lvlKpCount, int nmax, int ptsLdsCap, uint32_t levelMask, int level0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  DVS_CHAIN_PRIO();
  __shared__ int wsum[kOctTMax / 64 + 1];
  __shared__ int wsum2[2 * (kOctTMax / 64 + 1)];   // block_excl_scan_db
  int par = 0;
  __shared__ int s_n, s_c;
  __shared__ SortShared s_sort;
  const int tid = threadIdx.x;
  // grid = (frames, levels): the linear workgroup id is frame + frames x level, so the level-0 workgroups — the long ones — are
  // dispatched first and dealt round-robin over the XCDs, eight per XCD.  With (levels, frames) the id was level + 8 x frame and
  // XCD l received all 64 workgroups of level l: XCD 0 the 64 long ones, two per CU.
  // level0: first level of this launch — a launch may cover a class of levels only, with the node / point capacities (nmax, ptsLdsCap:
  // the dynamic LDS) of that class (graded launches of the quad-tree beside FAST, orb.hip)
  const int level = blockIdx.y + level0, f = blockIdx.x;
  if (tid >= OCT_T) return;  // wavefronts this level does not use leave before the first barrier
  if (!((levelMask >> level) & 1u)) {  // a level another rank owns: no keypoints from here
    if (tid == 0) { lvlKpCount[f * g->nlevels + level] = 0; candTotal[f * g->nlevels + level] = 0; }
    return;
  }
  const LevelGeom& L = g->lv[level];

  QtShared sh;
  {
    unsigned char* p = smem;
    sh.n0 = (QNode*)p; p += sizeof(QNode) * nmax;
    sh.n1 = (QNode*)p; p += sizeof(QNode) * nmax;
    sh.sortbuf = (unsigned long long*)p; p += 8 * nmax;
    sh.childCnt = (int*)p; p += 16 * nmax;
    sh.posArr = (int*)p; p += 4 * nmax;
    sh.flag = (int*)p; p += 4 * nmax;
    sh.expl = (int*)p; p += 4 * nmax;
    sh.ecum = (int*)p; p += 4 * nmax;
  }
  // point list + node-of-point: LDS when the level's candidates fit (the sweeps read them ~20 times), HBM otherwise
  uint32_t* ldsPts = (uint32_t*)(smem + (size_t)nmax * (2 * sizeof(QNode) + 8 + 16 + 4 * 4));
  int* ldsNodeOf = (int*)(ldsPts + ptsLdsCap);
  uint32_t* gpts = ptsAll + (uint64_t)f * g->ptsPerFrame + L.ptsOff;
  uint32_t* pts = gpts;
  int* nodeOf = nodeOfAll + (uint64_t)f * g->ptsPerFrame + L.ptsOff;
  const uint32_t* cnd = cand + (uint64_t)f * g->candPerFrame + L.candOff;
  const int* cc = cellCount + (uint64_t)f * g->totalCells + L.cellBase;
  int* co = cellOff + (uint64_t)f * g->totalCells + L.cellBase;

  QT_STAMP(0);
  // ---- gather: candidate order = cells row-major, pixels row-major inside a cell -------------
  {
    // cell offsets also live in LDS while they fit (the child-count array is dead until the roots exist): the gather below then has
    // ONE global round trip per candidate instead of two dependent ones
    int* coL = sh.childCnt;
    const bool coLds = L.nCells <= 4 * nmax;
    int carry = 0;
    int vnext = tid < L.nCells ? cc[tid] : 0;   // the next trip's count is requested before this trip's scan: one exposed global round trip, not one per trip
    for (int b = 0; b < L.nCells; b += OCT_T) {
      const int c = b + tid;
      const int v = vnext;
      vnext = c + OCT_T < L.nCells ? cc[c + OCT_T] : 0;
      int tot;
      const int ex = block_excl_scan_rt(v, wsum, tot);
      if (c < L.nCells) {
        co[c] = carry + ex;
        if (coLds) coL[c] = carry + ex;
      }
      carry += tot;
    }
    if (tid == 0) { s_n = carry; candTotal[f * g->nlevels + level] = carry; }
    __syncthreads();
    const bool inLds = s_n <= ptsLdsCap;
    if (inLds) { pts = ldsPts; nodeOf = ldsNodeOf; }
    // cell of candidate i: every cell stamps its index over its own slot range (cells hold a handful of candidates each, the
    // stores are fire-and-forget), then each candidate is one independent lookup.  The stamps live in the node-of-point array,
    // which is not in use yet.
    int* cellOf = inLds ? ldsNodeOf : nodeOf;
    const int total = s_n;
    for (int c = tid; c < L.nCells; c += OCT_T) {
      // a cell's count = the next cell's offset - its own (LDS) where the offsets are there: no second trip to the counts in HBM
      const int b0 = coLds ? coL[c] : co[c], b1 = coLds ? (c + 1 < L.nCells ? coL[c + 1] : total) : b0 + cc[c];
      for (int k = b0; k < b1; k++) cellOf[k] = c;
    }
    __syncthreads();
    // eight candidates per thread and trip: their loads are independent, so the global latency is paid once per trip (the level-0
    // workgroup walks 4000-6000 candidates with 256 threads: 16-24 dependent round trips one by one, 2-3 now)
    constexpr int kG = 8;
    for (int i0 = tid; i0 < total; i0 += kG * OCT_T) {
      int cidx[kG], off[kG];
      uint32_t v[kG];
#pragma unroll
      for (int u = 0; u < kG; u++) { const int i = i0 + u * OCT_T; cidx[u] = i < total ? cellOf[i] : 0; }
#pragma unroll
      for (int u = 0; u < kG; u++) off[u] = coLds ? coL[cidx[u]] : co[cidx[u]];
#pragma unroll
      for (int u = 0; u < kG; u++) { const int i = i0 + u * OCT_T; v[u] = i < total ? cnd[(uint64_t)cidx[u] * L.cellCap + (i - off[u])] : 0u; }
#pragma unroll
      for (int u = 0; u < kG; u++) {
        const int i = i0 + u * OCT_T;
        if (i < total) {
          gpts[i] = v[u];                 // kept in HBM too: dvs_orb_get_candidates reads it
          if (inLds) ldsPts[i] = v[u];
        }
      }
    }
    __syncthreads();
  }
  QT_STAMP(1);
  const int n = s_n;
  const int N = L.N;
  int cur = 0;
  int Scur = 0;   // length of the node list: uniform, kept in a register (every count below is a scan total all threads hold)

  // ---- roots (:559-601) ------------------------------------------------------------------------
  {
    const int nIni = L.nIni;
    for (int k = tid; k < nIni; k += OCT_T) sh.childCnt[k] = 0;
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 4 * OCT_T) {
      uint32_t p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = i0 + u * OCT_T; p[u] = i < n ? pts[i] : 0u; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * OCT_T;
        const bool v = i < n;
        const int r = v ? (int)__fdiv_rn((float)pt_x(p[u]), L.hX) : -1;  // vpIniNodes[kp.pt.x/hX]
        if (v) nodeOf[i] = r;  // root index for now
        // there are only round(w / h) roots (2 at 16:9): thousands of atomics on two addresses serialise, so a wavefront counts
        // its points per root with a ballot and adds once
        if (nIni <= 4) {
          for (int k = 0; k < nIni; k++) {
            const unsigned long long b = __ballot(r == k);
            if (lane_id() == 0 && b) atomicAdd(&sh.childCnt[k], __popcll(b));
          }
        } else if (v) {
          atomicAdd(&sh.childCnt[r], 1);
        }
      }
    }
    __syncthreads();
    int carry = 0;
    for (int b = 0; b < nIni; b += OCT_T) {
      const int k = b + tid;
      const int u = (k < nIni && sh.childCnt[k] > 0) ? 1 : 0;
      int tot;
      const int ex = block_excl_scan_rt(u, wsum, tot);
      if (k < nIni) sh.posArr[k] = u ? carry + ex : -1;
      if (u) {
        QNode nd;
        nd.ulx = (int16_t)(int)__fmul_rn(L.hX, (float)k);
        nd.brx = (int16_t)(int)__fmul_rn(L.hX, (float)(k + 1));
        nd.uly = 0; nd.bry = (int16_t)L.regionH;
        nd.cnt = sh.childCnt[k]; nd.pt = -1;
        sh.nodes_(0)[carry + ex] = nd;
      }
      carry += tot;
    }
    Scur = carry;   // (uniform: every thread holds the scan's total)
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 4 * OCT_T) {
      int r[4], pos[4], cnt[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = i0 + u * OCT_T; r[u] = i < n ? nodeOf[i] : 0; }
#pragma unroll
      for (int u = 0; u < 4; u++) pos[u] = sh.posArr[r[u]];
#pragma unroll
      for (int u = 0; u < 4; u++) cnt[u] = sh.nodes_(0)[max(pos[u], 0)].cnt;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * OCT_T;
        if (i < n) {
          if (cnt[u] == 1) { sh.nodes_(0)[pos[u]].pt = i; nodeOf[i] = -1; }
          else nodeOf[i] = pos[u];
        }
      }
    }
    __syncthreads();
  }
  // ---- main loop ---------------------------------------------------------------------------------
  // A sweep is a chain of short passes separated by workgroup barriers — ~14 of them in round 3's form, which were most of its time (the
  // level-0 tree: ~180 barriers in 56 us).  Now 6: the scans take one barrier instead of two (block_excl_scan_db), one scan carries the
  // children count AND the number of children to expand (packed 16 + 16 bits: both < 4 x 4096), the number of unsplit nodes is the total of
  // the scan qt_rebuild runs anyway, list length and totals live in registers (they are scan totals: uniform), and a thread reads flag /
  // posArr / ecum only of the nodes it wrote them for (k = tid, tid + OCT_T, ... in every loop).
  QT_STAMP(2);
  int qtp = 3;   // next stamp
  bool finish = (n == 0);
  while (!finish) {
    const int S = Scur;
    // full sweep: split every multi-point node (:622-681)
    qt_count_children(sh.nodes_(cur), S, sh.childCnt, (uint32_t*)sh.posArr, pts, nodeOf, n);
    QT_STAMP(qtp); qtp++;
    int T, nToExpand;
    {
      // children block: node k's children sit in front of the children of all earlier nodes
      int carry = 0;   // packed: children | children with more than one point << 16
      for (int b = 0; b < S; b += OCT_T) {
        const int k = b + tid;
        int e = 0, ne = 0;
        if (k < S) {
          const bool split = sh.nodes_(cur)[k].cnt > 1;
          sh.flag[k] = split ? 1 : 0;
          if (split) {
            e = __popc(qt_mask(sh.childCnt, k));
#pragma unroll
            for (int q = 0; q < 4; q++) ne += sh.childCnt[4 * k + q] > 1 ? 1 : 0;
          }
        }
        int tot;
        const int ex = block_excl_scan_db(e | (ne << 16), wsum2, par, tot);
        if (k < S) sh.ecum[k] = ((carry + ex) & 0xFFFF) + e;  // inclusive
        carry += tot;
      }
      T = carry & 0xFFFF; nToExpand = carry >> 16;
      for (int k = tid; k < S; k += OCT_T)
        if (sh.flag[k]) sh.posArr[k] = T - sh.ecum[k];
    }
    QT_STAMP(qtp); qtp++;
    const int nUnsplit = qt_rebuild(sh, cur, S, T, pts, nodeOf, n, wsum2, par);
    QT_STAMP(qtp); qtp++;
    cur ^= 1;
    const int Snew = T + nUnsplit;
    Scur = Snew;
    if (Snew >= N || Snew == S) { finish = true; break; }
    if (Snew + nToExpand * 3 <= N) continue;

    // ordered phase (:692-753)
    qtp = 40;
    QT_STAMP(qtp); qtp++;
    int m = qt_build_expand_list(sh, cur, T, wsum);
    while (!finish) {
      QT_STAMP(qtp); qtp++;
      const int Sb = Scur;
      if (m == 0) { finish = true; break; }  // nothing to split: size stays == prevSize
      qt_count_children(sh.nodes_(cur), Sb, sh.childCnt, (uint32_t*)sh.posArr, pts, nodeOf, n);
      for (int r = tid; r < m; r += OCT_T) {
        const QNode& nd = sh.nodes_(cur)[sh.expl[r]];
        sh.sortbuf[r] = ((unsigned long long)(uint32_t)nd.cnt << 28) | ((unsigned long long)(uint16_t)nd.ulx << 12) |
                        (unsigned long long)r;
      }
      for (int k = tid; k < Sb; k += OCT_T) sh.flag[k] = 0;
      __syncthreads();
      QT_STAMP(qtp); qtp++;
      qt_sort_block(sh.sortbuf, m, sh.ecum, sh.posArr, s_sort);  // ecum / posArr are dead until the sweep below
      QT_STAMP(qtp); qtp++;
      // processing order r' = 0..m-1 walks the sorted vector from the back (:701)
      int carry = 0;
      if (tid == 0) s_c = 0;
      __syncthreads();
      int below = 0;
      for (int b = 0; b < m; b += OCT_T) {
        const int r = b + tid;
        int e = 0;
        if (r < m) e = __popc(qt_mask(sh.childCnt, sh.expl[(int)(sh.sortbuf[m - 1 - r] & 0xFFFull)]));
        int tot;
        const int ex = block_excl_scan_rt(e, wsum, tot);
        if (r < m) {
          const int incl = carry + ex + e;
          sh.ecum[r] = incl;
          // list size after processing r'+1 nodes = Sb + sum(e - 1)
          if (Sb + incl - (r + 1) < N) below++;
        }
        carry += tot;
      }
      if (below) atomicAdd(&s_c, below);
      __syncthreads();
      const int c = s_c;
      const int M = c < m ? c + 1 : m;  // break right after the split that reaches N (:746-747)
      const int Tm = sh.ecum[M - 1];
      __syncthreads();
      for (int r = tid; r < M; r += OCT_T) {
        const int k = sh.expl[(int)(sh.sortbuf[m - 1 - r] & 0xFFFull)];
        sh.flag[k] = 1;
        sh.posArr[k] = Tm - sh.ecum[r];
      }
      __syncthreads();
      QT_STAMP(qtp); qtp++;
      (void)qt_rebuild(sh, cur, Sb, Tm, pts, nodeOf, n, wsum2, par);
      QT_STAMP(qtp); qtp++;
      cur ^= 1;
      const int Sn = Tm + (Sb - M);
      Scur = Sn;
      if (Sn >= N || Sn == Sb) { finish = true; break; }
      m = qt_build_expand_list(sh, cur, Tm, wsum);
    }
  }
  // ---- best point per node (:757-776), list order = output order ----------------------------------
  QT_STAMP(90);
  const int S = (n == 0) ? 0 : Scur;
  QNode* nodes = sh.nodes_(cur);
  int* best = sh.childCnt;
  for (int k = tid; k < S; k += OCT_T) best[k] = 0;
  __syncthreads();
  for (int i0 = tid; i0 < n; i0 += 4 * OCT_T) {
    int k[4];
    uint32_t p[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = i0 + u * OCT_T; k[u] = i < n ? nodeOf[i] : -1; p[u] = i < n ? pts[i] : 0u; }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (k[u] >= 0) atomicMax((unsigned int*)&best[k[u]], ((uint32_t)pt_s(p[u]) << 24) | (0xFFFFFFu - (uint32_t)(i0 + u * OCT_T)));
  }
  __syncthreads();
  uint32_t* outp = lvlKp + (uint64_t)f * g->kpBlock + L.kpOff;
  for (int k = tid; k < S; k += OCT_T) {
    const QNode nd = nodes[k];
    const int i = nd.cnt == 1 ? nd.pt : (int)(0xFFFFFFu - ((uint32_t)best[k] & 0xFFFFFFu));
    if (k < N + 4) outp[k] = pts[i];
  }
  if (tid == 0) lvlKpCount[f * g->nlevels + level] = min(S, N + 4);
  QT_STAMP(91);
#ifdef DVS_QT_PROF
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { g_qt_prof[92] = (unsigned long long)n; g_qt_prof[93] = (unsigned long long)S; g_qt_prof[94] = (unsigned long long)qtp; }
#endif
}

__global__ __launch_bounds__(kOctTMax) void k_octree(const Geom* __restrict__ g, const uint32_t* __restrict__ cand,
                                                const int* __restrict__ cellCount, int* __restrict__ cellOff,
                                                uint32_t* __restrict__ ptsAll, int* __restrict__ nodeOfAll,
                                                int* __restrict__ candTotal, uint32_t* __restrict__ lvlKp,
                                                int* __restrict__ lvlKpCount, int nmax, int ptsLdsCap, uint32_t levelMask, int level0) {
#ifdef DVS_EXP_OCT_PRIO   /* EXPERIMENT (round 5): issue priority of the latency-bound trees over what runs beside them */
  __builtin_amdgcn_s_setprio(DVS_EXP_OCT_PRIO);
#endif
  octree_body(g, cand, cellCount, cellOff, ptsAll, nodeOfAll, candTotal, lvlKp, lvlKpCount, nmax, ptsLdsCap, levelMask, level0);
}

#undef OCT_T
// =============================================================================================
// 7x7 Gaussian, sigma 2, BORDER_REFLECT_101 on the level itself — OpenCV's 8-bit fixed-point path:
// horizontal Q8.8 (exact in u16), vertical Q16.16, (acc + 32768) >> 16.  Tile = 64 x 16 outputs.
// =============================================================================================
__global__ __launch_bounds__(256) void k_blur(const Geom* __restrict__ g, const BlurTile* __restrict__ tiles, ImgSrc src,
                                              u8* __restrict__ blur) {
  __shared__ u8 in[22][72];
  __shared__ uint16_t hb[22][64];
  const BlurTile t = tiles[blockIdx.x];
  const int f = blockIdx.y;
  if (!((src.levelMask >> t.level) & 1u)) return;
  const LevelGeom& L = g->lv[t.level];
  int pitch;
  const u8* img = level_ptr(g, src, f, t.level, pitch);
  const int x0 = t.tx * 64, y0 = t.ty * 16;
  const int tid = threadIdx.x;
  for (int p = tid; p < 22 * 70; p += 256) {
    const int r = p / 70, c = p - r * 70;
    const int sy = reflect101(y0 + r - 3, L.h), sx = reflect101(x0 + c - 3, L.w);
    in[r][c] = img[(uint64_t)sy * pitch + sx];
  }
  __syncthreads();
  const int k0 = g->gk[0], k1 = g->gk[1], k2 = g->gk[2], k3 = g->gk[3], k4 = g->gk[4], k5 = g->gk[5], k6 = g->gk[6];
  for (int p = tid; p < 22 * 64; p += 256) {
    const int r = p >> 6, c = p & 63;
    const u8* s = &in[r][c];
    hb[r][c] = (uint16_t)(k0 * s[0] + k1 * s[1] + k2 * s[2] + k3 * s[3] + k4 * s[4] + k5 * s[5] + k6 * s[6]);
  }
  __syncthreads();
  u8* dst = blur + (uint64_t)f * g->frameBytes + L.off;
  const int c = tid & 63;
  const int rq = tid >> 6;
  if (x0 + c < L.w) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = rq * 4 + i;
      if (y0 + r < L.h) {
        const uint32_t acc = (uint32_t)k0 * hb[r][c] + (uint32_t)k1 * hb[r + 1][c] + (uint32_t)k2 * hb[r + 2][c] +
                             (uint32_t)k3 * hb[r + 3][c] + (uint32_t)k4 * hb[r + 4][c] + (uint32_t)k5 * hb[r + 5][c] +
                             (uint32_t)k6 * hb[r + 6][c];
        dst[(uint64_t)(y0 + r) * L.pitch + x0 + c] = (u8)((acc + 32768u) >> 16);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming variant of the same filter (used whenever rows are 4-byte aligned): no LDS.  One wavefront
// owns a strip of <= 256 columns (4 per lane, one aligned dword) and walks kBlurBand rows downwards:
//   horizontal: the 12-byte window [left | own | right] comes from the neighbour lanes; every output
//               is two v_dot4_u32_u8 against the packed kernel bytes (exact: Q8.8 sum <= 65 280)
//   vertical  : the last 7 horizontal rows live in registers (ring of 7 x 4 values); one output row per
//               step with v_mad_u32_u24 in Q16.16, (acc + 32768) >> 16, one coalesced dword store.
// Image edges use BORDER_REFLECT_101 on the level itself: rows by (wave-uniform) index reflection,
// columns by a per-byte reflected gather that only the first / last lane of an edge strip executes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t blur_word_reflect(const u8* row, int xw, int w) {
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) v |= (uint32_t)row[reflect101(xw + i, w)] << (8 * i);
  return v;
}

// one wavefront = one strip `wi` of frame `f` (no LDS, no workgroup barrier: k_octree_blur runs it in workgroups beside the quad-trees')
__device__ __forceinline__ void blur_stream_wave(const Geom* __restrict__ g, const BlurStrip* __restrict__ strips, int nstrips,
                                                 const ImgSrc& src, u8* __restrict__ blur, int wi, int f) {
  if (wi >= nstrips) return;
  const BlurStrip s = strips[wi];
  if (!((src.levelMask >> s.level) & 1u)) return;
  const int lane = lane_id();
  const LevelGeom& L = g->lv[s.level];
  int pitch;
  const u8* img = level_ptr(g, src, f, s.level, pitch);
  u8* dst = blur + (uint64_t)f * g->frameBytes + L.off;
  const int W = L.w, H = L.h;
  const int nlo = (s.w + 3) >> 2;          // output lanes 1..nlo; lanes 0 and nlo+1 only supply the halo words
  const int x = s.x0 - 4 + lane * 4;       // first pixel of this lane's word
  const bool outl = lane >= 1 && lane <= nlo;
  // Columns: levels >= 1 carry >= 8 reflected columns right of the image (written by k_resize4), so plain loads work
  // there; the left border of every level and the right border of level 0 (width % 4 == 0, no padding in a caller's
  // buffer) are byte permutations of the edge lane's own word.  No branches, no byte gathers in the row loop.
  const int xmax = s.level == 0 ? W - 4 : ((W + 8) & ~3) - 4;  // last loadable word start (levels >= 1: padded width)
  const int xl = min(max(x, 0), xmax);
  const uint32_t q0 = (uint32_t)g->gk[0], q1 = (uint32_t)g->gk[1], q2 = (uint32_t)g->gk[2], q3 = (uint32_t)g->gk[3], q4 = (uint32_t)g->gk[4],
                 q5 = (uint32_t)g->gk[5], q6 = (uint32_t)g->gk[6];
  // BORDER_REFLECT_101 in the columns is folded into the WEIGHTS of the one lane that holds a border word (per-lane registers, set
  // once): px -3, -2, -1 = px 3, 2, 1 add their taps to the own word's bytes and the left word gets weight 0; likewise px W, W + 1,
  // W + 2 = px W - 2, W - 3, W - 4 at the right border of level 0 (levels >= 1 read the mirrored columns k_resize4 wrote).  Sums of
  // two Q8 weights stay below 256.  (Round 2a built the mirrored neighbour word per row: two permutes and two selects.)
  uint32_t wL0 = q0 << 8 | q1 << 16 | q2 << 24, wO0 = q3 | q4 << 8 | q5 << 16 | q6 << 24;
  uint32_t wL1 = q0 << 16 | q1 << 24, wO1 = q2 | q3 << 8 | q4 << 16 | q5 << 24, wR1 = q6;
  uint32_t wL2 = q0 << 24, wO2 = q1 | q2 << 8 | q3 << 16 | q4 << 24, wR2 = q5 | q6 << 8;
  uint32_t wO3 = q0 | q1 << 8 | q2 << 16 | q3 << 24, wR3 = q4 | q5 << 8 | q6 << 16;
  if (x == 0) {
    wL0 = 0; wO0 = q3 | (q2 + q4) << 8 | (q1 + q5) << 16 | (q0 + q6) << 24;
    wL1 = 0; wO1 = q2 | (q1 + q3) << 8 | (q0 + q4) << 16 | q5 << 24;
    wL2 = 0; wO2 = q1 | (q0 + q2) << 8 | q3 << 16 | q4 << 24;
  }
  if (s.level == 0 && x + 4 == W) {
    wR1 = 0; wO1 = q2 | q3 << 8 | (q4 + q6) << 16 | q5 << 24;
    wR2 = 0; wO2 = q1 | (q2 + q6) << 8 | (q3 + q5) << 16 | q4 << 24;
    wR3 = 0; wO3 = (q0 + q6) | (q1 + q5) << 8 | (q2 + q4) << 16 | q3 << 24;
  }
  const uint32_t kv[7] = {(uint32_t)g->gk[0], (uint32_t)g->gk[1], (uint32_t)g->gk[2], (uint32_t)g->gk[3],
                          (uint32_t)g->gk[4], (uint32_t)g->gk[5], (uint32_t)g->gk[6]};
  const int rows = min(g->blurBand, H - s.y0);
  const int T = rows + 6;

  auto load_own = [&](int k) -> uint32_t {
    int sy = s.y0 + k - 3;                 // BORDER_REFLECT_101 on rows (H >= 7 always holds for a level)
    sy = sy < 0 ? -sy : sy;
    sy = sy >= H ? 2 * H - 2 - sy : sy;
    // wave-uniform row offset (SALU) + per-lane column as ONE 32-bit offset from the uniform level base: the 64-bit form compiled
    // to a quarter-rate v_mad_u64_u32 per load
    return *reinterpret_cast<const uint32_t*>(img + (uint32_t)(sy * pitch + xl));
  };
  typedef unsigned short us2v __attribute__((ext_vector_type(2)));
  const us2v k01 = {(unsigned short)kv[0], (unsigned short)kv[1]}, k23 = {(unsigned short)kv[2], (unsigned short)kv[3]},
             k45 = {(unsigned short)kv[4], (unsigned short)kv[5]};
  const uint32_t kv6s = __builtin_amdgcn_readfirstlane(kv[6]);   // the column filter's last weight, pinned to an SGPR (mad_u24_s)
  uint32_t ring[7][4], hprev[4] = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 7; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) ring[a][b] = 0;

  // rows are fetched a whole group of 7 ahead: 7..14 independent 256-byte row loads in flight per wavefront keep
  // enough bytes outstanding for HBM latency (with 1-2 in flight the kernel ran latency-bound at ~1.5 TB/s)
  uint32_t cur[7], nxt[7];
#pragma unroll
  for (int kk = 0; kk < 7; kk++) cur[kk] = kk < T ? load_own(kk) : 0u;
  for (int k0 = 0; k0 < T; k0 += 7) {
#pragma unroll
    for (int kk = 0; kk < 7; kk++) nxt[kk] = (k0 + 7 + kk < T) ? load_own(k0 + 7 + kk) : 0u;
#pragma unroll
    for (int kk = 0; kk < 7; kk++) {
      const int k = k0 + kk;
      if (k < T) {
        const uint32_t own = cur[kk];
        // neighbour words by DPP whole-wave shifts (wave_shr:1 / wave_shl:1: one VALU move each, no LDS crossbar trip;
        // semantics checked on gfx950: lane i receives lane i-1 / i+1, the open end keeps `old` = 0)
        // bound_ctrl: the lane at the open end reads 0 without a destination initialised first (one v_mov per shift less)
        uint32_t left = __builtin_amdgcn_update_dpp(0u, own, 0x138, 0xf, 0xf, true);
        uint32_t right = __builtin_amdgcn_update_dpp(0u, own, 0x130, 0xf, 0xf, true);
        // px -3,-2,-1 = px 3,2,1 ; px W,W+1,W+2 = px W-2,W-3,W-4
        // px x+j takes bytes x+j-3 .. x+j+3 of the 12-byte window (left | own | right): instead of shifting the DATA to a common
        // alignment (six v_alignbyte per row) the WEIGHTS are kept in the ten alignments the four pixels need (scalar constants):
        // 2 + 3 + 3 + 2 v_dot4_u32_u8
        uint32_t hc[4];
        hc[0] = __builtin_amdgcn_udot4(own, wO0, __builtin_amdgcn_udot4(left, wL0, 0u, false), false);
        hc[1] = __builtin_amdgcn_udot4(right, wR1, __builtin_amdgcn_udot4(own, wO1, __builtin_amdgcn_udot4(left, wL1, 0u, false), false), false);
        hc[2] = __builtin_amdgcn_udot4(right, wR2, __builtin_amdgcn_udot4(own, wO2, __builtin_amdgcn_udot4(left, wL2, 0u, false), false), false);
        hc[3] = __builtin_amdgcn_udot4(right, wR3, __builtin_amdgcn_udot4(own, wO3, 0u, false), false);
        // ring slot kk holds the PAIR (row k-1, row k) of horizontal sums as two u16 halves: the 7-tap column filter is then
        // three v_dot2_u32_u16 on the pairs formed at rows k-5, k-3, k-1 plus one multiply-add for row k
#pragma unroll
        for (int j = 0; j < 4; j++) { ring[kk][j] = hprev[j] | (hc[j] << 16); hprev[j] = hc[j]; }
        if (k >= 6 && outl) {
          uint32_t acc[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            acc[j] = mad_u24_s(hc[j], kv6s, 32768u);
            acc[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, ring[(kk + 2) % 7][j]), k01, acc[j], false);  // rows k-6, k-5
            acc[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, ring[(kk + 4) % 7][j]), k23, acc[j], false);  // rows k-4, k-3
            acc[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2v, ring[(kk + 6) % 7][j]), k45, acc[j], false);  // rows k-2, k-1
          }
          // byte 2 of each Q16.16 sum is the output pixel: three byte permutes gather the four of them (were four extracts + four inserts)
          const uint32_t o = __builtin_amdgcn_perm(__builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0602u), __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u),
                                                   0x05040100u);
          u8* orow = dst + (uint32_t)((s.y0 + k - 6) * L.pitch + x);
          *reinterpret_cast<uint32_t*>(orow) = o;  // the blurred block has the padded pitch too: a tail word may spill into it
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 7; kk++) cur[kk] = nxt[kk];
  }
}

__global__ __launch_bounds__(256) void k_blur_stream(const Geom* __restrict__ g, const BlurStrip* __restrict__ strips, int nstrips,
                                                     ImgSrc src, u8* __restrict__ blur) {
  // wave-uniform strip index: strip geometry in SGPRs
  blur_stream_wave(g, strips, nstrips, src, blur, blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), blockIdx.y);
}

// Quad-trees and the streaming blur in ONE launch, for a lane of dvs_pipeline (one stream, a few frames): there the blur (9 us for one
// frame) waits behind the latency-bound trees (54 us) although neither needs the other.  Grid (frames, levels + blur rows): rows
// [0, treeLevels) are the trees' workgroups, the others run eight blur strips each — one per wavefront, no LDS, no barrier.
__global__ __launch_bounds__(kOctTMax) void k_octree_blur(const Geom* __restrict__ g, const uint32_t* __restrict__ cand,
                                                     const int* __restrict__ cellCount, int* __restrict__ cellOff,
                                                     uint32_t* __restrict__ ptsAll, int* __restrict__ nodeOfAll,
                                                     int* __restrict__ candTotal, uint32_t* __restrict__ lvlKp,
                                                     int* __restrict__ lvlKpCount, int nmax, int ptsLdsCap, uint32_t levelMask, int treeLevels,
                                                     const BlurStrip* __restrict__ strips, int nstrips, ImgSrc src, u8* __restrict__ blur) {
  if ((int)blockIdx.y >= treeLevels) {
    const int wpb = (int)blockDim.x >> 6;
    blur_stream_wave(g, strips, nstrips, src, blur, ((int)blockIdx.y - treeLevels) * wpb + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), blockIdx.x);
    return;
  }
  octree_body(g, cand, cellCount, cellOff, ptsAll, nodeOfAll, candTotal, lvlKp, lvlKpCount, nmax, ptsLdsCap, levelMask, 0);
}

// ---------------------------------------------------------------------------------------------
// Matrix-core form of the same blur (round 2).  The 7 Q8 weights (18 .. 56) fit int8, so both passes are banded (Toeplitz)
// int8 products on v_mfma_i32_32x32x32_i8 with exact int32 accumulation — the pipe every other kernel of the path leaves idle —
// and the VALU keeps only the byte packing between them: 0.07 wave-instructions per pixel instead of 0.29.
//   horizontal  h[r][c] = sum_t k_t p[r][c + t - 3]          A = 32 rows x 64 source columns of (p - 128)   (two 16-byte loads per lane)
//                                                             B = the strip's 64 x 32 band matrix (host table; REFLECT_101 at the
//                                                                 left / right border is folded into its weights)
//               acc = h - 32768, so byte 1 of acc is (h >> 8) - 128 and byte 0 is h & 255: the two int8 operands of pass two.
//   vertical    V = sum_t k_t h[r + t - 3] = 256 (S_hi + 32768) + (S_lo + 32768)   with S_hi / S_lo the band products of the
//               high / low bytes;  out = (V + 32768) >> 16 = (S_hi + 32768 + ((S_lo + 65536) >> 8)) >> 8  (nested floors are
//               exact), i.e. the low chain starts from the constant 65536 + (32768 << 8) and its >> 8 is the high chain's C input.
//               A tile of 32 output rows takes the 32 h-rows from 3 above it (the carried block) and the block after it.
//   rows outside the image are fetched from their REFLECT_101 source row, so the vertical band matrix is always the interior one.
// A lane's accumulators of pass one (column = lane & 31, rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) are exactly the 16 k-values
// of one operand fragment of pass two once the band matrix' k order is permuted the same way (host table), so no data moves between
// lanes.  Pass two runs transposed (A = h bytes, B = band matrix): a lane then holds 4 x 4 consecutive output columns of one row.
// Same results as k_blur_stream bit for bit (integer arithmetic throughout).
// ---------------------------------------------------------------------------------------------
typedef int bv4i __attribute__((ext_vector_type(4)));
typedef int bv16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_blur_mfma(const Geom* __restrict__ g, const BlurCol* __restrict__ items, int nitems, ImgSrc src,
                                                   u8* __restrict__ blur, const uint4* __restrict__ tab, int avt) {
  // A workgroup filters a 128-column super-strip: wavefront w owns the 32-column strip w of it.  LDS exists for the memory side
  // only — a lane of an operand fragment owns one ROW, so fragment-shaped global accesses touch 32 different lines per instruction
  // (measured: 0.25 ms against 0.04 ms of arithmetic).  Through LDS, ten consecutive lanes fetch the 160 contiguous source bytes
  // of a row once for all four strips (LDS-DMA, two blocks in flight) and eight consecutive lanes store the 128 contiguous bytes
  // of an output row.
  constexpr int kRowB = 160, kBlkB = 32 * kRowB;     // source block: 32 rows x (16 + 128 + 16) bytes
  __shared__ __attribute__((aligned(16))) unsigned char sblk[2][kBlkB];
  __shared__ __attribute__((aligned(16))) unsigned char sout[2][32 * 128];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xid = xcd_contiguous_id();            // frames -> XCDs contiguously: neighbouring workgroups share one L2
  const int f = xid / (int)gridDim.x;
  const BlurCol it = items[xid - f * (int)gridDim.x];
  if (!((src.levelMask >> it.level) & 1u)) return;
  const int lane = lane_id(), m = lane & 31, half = lane >> 5;
  const LevelGeom& L = g->lv[it.level];
  int pitch;
  const u8* img = level_ptr(g, src, f, it.level, pitch);
  u8* dst = blur + (uint64_t)f * g->frameBytes + L.off;
  const int W = L.w, H = L.h, dp = L.pitch;
  const int C0 = it.strip * 128;                  // first column of the super-strip
  const int ti = (it.tab + wv) * 2;
  const bv4i Bh1 = __builtin_bit_cast(bv4i, tab[(ti + 0) * 64 + lane]), Bh2 = __builtin_bit_cast(bv4i, tab[(ti + 1) * 64 + lane]);
  const bv4i Av0 = __builtin_bit_cast(bv4i, tab[(avt * 2 + 0) * 64 + lane]), Av1 = __builtin_bit_cast(bv4i, tab[(avt * 2 + 1) * 64 + lane]);
  const bv4i bias = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
  const bv16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr uint32_t kC = 65536u + (32768u << 8);

  // DMA side: wavefront w fetches rows 8 w .. 8 w + 7 of a block = 80 consecutive 16-byte slots (64 + 16 lanes).  Chunks outside
  // the row carry zero weights and are fetched from inside it.
  const int sA = lane, sB = 64 + (lane & 15);
  const int rA = 8 * wv + sA / 10, rB = 8 * wv + sB / 10;
  const uint32_t cA = (uint32_t)min(max(C0 - 16 + 16 * (sA % 10), 0), pitch - 16), cB = (uint32_t)min(max(C0 - 16 + 16 * (sB % 10), 0), pitch - 16);
  auto src_off = [&](int vr, uint32_t col) -> uint32_t {   // virtual row -> REFLECT_101 source row (clamped: rows further out than
    int r = vr < 0 ? -vr : vr;                             // the 3-row border only feed rows that are not stored)
    r = r >= H ? 2 * H - 2 - r : r;
    r = min(max(r, 0), H - 1);
    return mad_u24((uint32_t)r, (uint32_t)pitch, col);
  };
  auto fetch = [&](int b, unsigned char* buf) {   // h-block b = source rows 32 b - 3 .. 32 b + 28
    unsigned char* wb = buf + wv * (80 * 16);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + src_off(32 * b - 3 + rA, cA)),
                                     (__attribute__((address_space(3))) void*)wb, 16, 0, 0);
    if (lane < 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + src_off(32 * b - 3 + rB, cB)),
                                       (__attribute__((address_space(3))) void*)(wb + 1024), 16, 0, 0);
  };
  // fragment side: lane (m, half) of wavefront w holds source chunks 2 w + half and 2 w + 2 + half of row m
  const int fo1 = m * kRowB + 16 * (2 * wv + half), fo2 = fo1 + 32;
  auto hblock = [&](const unsigned char* buf, bv4i& hi, bv4i& lo) {
    const bv4i a1 = *reinterpret_cast<const bv4i*>(buf + fo1) ^ bias;
    const bv4i a2 = *reinterpret_cast<const bv4i*>(buf + fo2) ^ bias;
    bv16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, Bh1, zero, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, Bh2, acc, 0, 0, 0);
    uint32_t P[8];
#pragma unroll
    for (int q = 0; q < 8; q++) P[q] = __builtin_amdgcn_perm((uint32_t)acc[2 * q + 1], (uint32_t)acc[2 * q], 0x05040100u);
#pragma unroll
    for (int d = 0; d < 4; d++) {
      hi[d] = (int)__builtin_amdgcn_perm(P[2 * d + 1], P[2 * d], 0x07050301u);
      lo[d] = (int)(__builtin_amdgcn_perm(P[2 * d + 1], P[2 * d], 0x06040200u) ^ 0x80808080u);
    }
  };
  const int wo = m * 128 + 32 * wv + 4 * half;                               // + 8 d: this lane's 4 x 4 output columns of row m
  const int orow = (int)threadIdx.x >> 3, ocol = C0 + 16 * ((int)threadIdx.x & 7);   // store side: 8 lanes per 128-byte output row
  uint32_t so = mad_u24((uint32_t)(32 * it.t0 + orow), (uint32_t)dp, (uint32_t)ocol);
  auto vtile = [&](int t, const bv4i& chi, const bv4i& clo, const bv4i& nhi, const bv4i& nlo) {
    bv16i lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(clo, Av0, zero, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(nlo, Av1, lo, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) lo[k] = (int)(((uint32_t)lo[k] + kC) >> 8);   // only bits 8 .. 23 of the sum reach the output
    bv16i u = __builtin_amdgcn_mfma_i32_32x32x32_i8(chi, Av0, lo, 0, 0, 0);
    u = __builtin_amdgcn_mfma_i32_32x32x32_i8(nhi, Av1, u, 0, 0, 0);
    unsigned char* ot = sout[t & 1];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)u[4 * d + 1], (uint32_t)u[4 * d], 0x0c0c0501u);
      const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)u[4 * d + 3], (uint32_t)u[4 * d + 2], 0x0c0c0501u);
      *reinterpret_cast<uint32_t*>(ot + wo + 8 * d) = p01 | (p23 << 16);
    }
    __syncthreads();
    const bv4i o = *reinterpret_cast<const bv4i*>(ot + threadIdx.x * 16);
    // columns past the width land in the row's padding (pitch >= w + 8 rounded up to 64); whole 16-byte groups past it are skipped
    if (32 * t + orow < H && ocol < W) *reinterpret_cast<bv4i*>(dst + so) = o;
    so += 32u * (uint32_t)dp;
  };

  bv4i hiA, loA, hiB, loB;
  fetch(it.t0, sblk[0]);
  fetch(it.t0 + 1, sblk[1]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  hblock(sblk[0], hiA, loA);
  const int tend = it.t0 + it.nt;
  // two tiles per trip (the carried block alternates between the register sets).  Per tile: every wavefront waits for its own DMA,
  // the barrier makes the block complete, the fragments are read, and the buffer two blocks back — read by everybody before
  // this barrier — takes the next DMA, which lands while the tile is computed.
  for (int t = it.t0; t < tend; t += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    hblock(sblk[1], hiB, loB);              // block t + 1
    if (t + 1 < tend) fetch(t + 2, sblk[0]);
    vtile(t, hiA, loA, hiB, loB);
    if (t + 1 < tend) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      hblock(sblk[0], hiA, loA);            // block t + 2
      if (t + 2 < tend) fetch(t + 3, sblk[1]);
      vtile(t + 1, hiB, loB, hiA, loA);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The matrix-core blur WITHOUT LDS (round 3, DVS_BLUR_MFMA=2): every wavefront filters its own 32-column strip, operand fragments
// come straight from global memory (a lane = one row: 16-byte pieces of 32 different rows per load instruction) and the output
// leaves as one 16-byte store per lane (two v_permlane32_swap put a lane's 4 x 4 columns side by side).  No barriers, no LDS: the
// kernel can share a CU with FAST or the quad-tree, whose workgroups take all of its LDS.  Same tables, same arithmetic, same
// results as k_blur_mfma.  Alone 0.115 ms per 64 frames (k_blur_stream 0.081, k_blur_mfma 0.118); in the step it LOSES in either
// place — beside the quad-tree 0.511 against 0.482 ms (the quad-tree stretches 0.099 -> 0.141), started ahead of FAST 0.497 against
// 0.473 (FAST 0.352 -> 0.340, but the level chain 0.213 -> 0.292 and the quad-tree 0.098 -> 0.119): DESIGN.md 4b.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur_mfma_direct(const Geom* __restrict__ g, const BlurCol* __restrict__ items, int nitems, ImgSrc src,
                                                          u8* __restrict__ blur, const uint4* __restrict__ tab, int avt) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xid = xcd_contiguous_id();
  const int f = xid / (int)gridDim.x;
  const BlurCol it = items[xid - f * (int)gridDim.x];
  if (!((src.levelMask >> it.level) & 1u)) return;
  const int lane = lane_id(), m = lane & 31, half = lane >> 5;
  const LevelGeom& L = g->lv[it.level];
  int pitch;
  const u8* img = level_ptr(g, src, f, it.level, pitch);
  u8* dst = blur + (uint64_t)f * g->frameBytes + L.off;
  const int W = L.w, H = L.h, dp = L.pitch;
  const int c0 = it.strip * 128 + 32 * wv;        // this wavefront's strip
  if (c0 >= W) return;
  const int ti = (it.tab + wv) * 2;
  const bv4i Bh1 = __builtin_bit_cast(bv4i, tab[(ti + 0) * 64 + lane]), Bh2 = __builtin_bit_cast(bv4i, tab[(ti + 1) * 64 + lane]);
  const bv4i Av0 = __builtin_bit_cast(bv4i, tab[(avt * 2 + 0) * 64 + lane]), Av1 = __builtin_bit_cast(bv4i, tab[(avt * 2 + 1) * 64 + lane]);
  const bv4i bias = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
  const bv16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr uint32_t kC = 65536u + (32768u << 8);
  // source pieces of this lane: columns c0 - 16 + 16 half and 32 further; pieces outside the row carry zero weights and are fetched from inside it
  const uint32_t col1 = (uint32_t)min(max(c0 - 16 + 16 * half, 0), pitch - 16), col2 = (uint32_t)min(max(c0 + 16 + 16 * half, 0), pitch - 16);
  auto row_off = [&](int vr) -> uint32_t {
    int r = vr < 0 ? -vr : vr;
    r = r >= H ? 2 * H - 2 - r : r;
    r = min(max(r, 0), H - 1);
    return (uint32_t)r * (uint32_t)pitch;
  };
  bv4i r1, r2;
  auto issue = [&](int b) {   // h-block b = source rows 32 b - 3 .. 32 b + 28
    const uint32_t ro = row_off(32 * b - 3 + m);
    r1 = *reinterpret_cast<const bv4i*>(img + ro + col1);
    r2 = *reinterpret_cast<const bv4i*>(img + ro + col2);
  };
  auto hcomp = [&](const bv4i& x1, const bv4i& x2, bv4i& hi, bv4i& lo) {
    bv16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x1 ^ bias, Bh1, zero, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x2 ^ bias, Bh2, acc, 0, 0, 0);
    uint32_t P[8];
#pragma unroll
    for (int q = 0; q < 8; q++) P[q] = __builtin_amdgcn_perm((uint32_t)acc[2 * q + 1], (uint32_t)acc[2 * q], 0x05040100u);
#pragma unroll
    for (int d = 0; d < 4; d++) {
      hi[d] = (int)__builtin_amdgcn_perm(P[2 * d + 1], P[2 * d], 0x07050301u);
      lo[d] = (int)(__builtin_amdgcn_perm(P[2 * d + 1], P[2 * d], 0x06040200u) ^ 0x80808080u);
    }
  };
  const int ocol = c0 + 16 * half;
  uint32_t so = (uint32_t)(32 * it.t0 + m) * (uint32_t)dp + (uint32_t)ocol;
  auto vtile = [&](int t, const bv4i& chi, const bv4i& clo, const bv4i& nhi, const bv4i& nlo) {
    bv16i lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(clo, Av0, zero, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(nlo, Av1, lo, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) lo[k] = (int)(((uint32_t)lo[k] + kC) >> 8);
    bv16i u = __builtin_amdgcn_mfma_i32_32x32x32_i8(chi, Av0, lo, 0, 0, 0);
    u = __builtin_amdgcn_mfma_i32_32x32x32_i8(nhi, Av1, u, 0, 0, 0);
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)u[4 * d + 1], (uint32_t)u[4 * d], 0x0c0c0501u);
      const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)u[4 * d + 3], (uint32_t)u[4 * d + 2], 0x0c0c0501u);
      o[d] = p01 | (p23 << 16);   // columns c0 + 4 half + 8 d .. + 3 of row m
    }
    // lower lanes hold columns {0, 8, 16, 24}, upper lanes {4, 12, 20, 28} (+ 0..3): after exchanging the upper half of o[0] / o[1] with the
    // lower half of o[2] / o[3], (o0, o2, o1, o3) are 16 consecutive columns — 0..15 in the lower lanes, 16..31 in the upper ones
    const auto s02 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
    if (32 * t + m < H && ocol < W) *reinterpret_cast<uint4*>(dst + so) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
    so += 32u * (uint32_t)dp;
  };
  bv4i hiA, loA, hiB, loB;
  issue(it.t0);
  bv4i x1 = r1, x2 = r2;
  issue(it.t0 + 1);
  hcomp(x1, x2, hiA, loA);
  const int tend = it.t0 + it.nt;
  for (int t = it.t0; t < tend; t += 2) {
    x1 = r1; x2 = r2;
    if (t + 1 < tend) issue(t + 2);
    hcomp(x1, x2, hiB, loB);                // block t + 1
    vtile(t, hiA, loA, hiB, loB);
    if (t + 1 < tend) {
      x1 = r1; x2 = r2;
      if (t + 2 < tend) issue(t + 3);
      hcomp(x1, x2, hiA, loA);              // block t + 2
      vtile(t + 1, hiB, loB, hiA, loA);
    }
  }
}

// =============================================================================================
// orientation + descriptor + final keypoint record.  One wavefront per keypoint.
// =============================================================================================
// Eight keypoint slots per wavefront (fixed (level, index) slots of the per-level keypoint block).
//  * slot -> (level, index, output position), the keypoint word and its addresses are resolved by lanes 0..7 in one pass and
//    broadcast with v_readlane;
//  * orientation: one unaligned 16-byte load per lane and keypoint (lane = 2*row + half of the 31 x 31 patch), all eight in
//    flight together; the 16 moment sums are reduced with a transposing butterfly (17 shuffles instead of 96) that leaves
//    keypoint k's totals in lanes 4k..4k+3, where fastAtan2 and the (double precision, glibc-exact) sin/cos run ONCE for all
//    eight keypoints;
//  * BRIEF: the 512 samples are NOT gathered from HBM/L2 — a fully divergent byte gather costs the CU's texture-address unit one
//    cache line per lane.  The 37 x 37 blurred window (every sample lies within +-18 px) is staged into LDS with two coalesced
//    16-byte-per-lane loads (3 lanes per row, next keypoint's loads issued behind the current keypoint's sampling), and the
//    samples are LDS byte reads.  The pattern is unpacked to floats once per wave.
constexpr int kDescKP = 8;
constexpr int kWinR = 18, kWinRows = 2 * kWinR + 1, kWinPitch = 48;
// (Round 1 split the kernel at the angle so that the orientation half could run beside the blur; with the deferred descriptor
// stage of the pipelined schedule that bought nothing — 0.686 vs 0.681 ms per step — and the split was removed.)
// Spatial visiting order for the descriptor stage (round 4): the quad-tree leaves a level's keypoints in LIST order (reverse creation
// order of the nodes, ORBextractor.cpp:639-674), i.e. scattered over the level, and a wavefront of k_describe takes eight consecutive
// ones — eight 37 x 48-byte windows and eight 31 x 31 patches in eight different places: 382 MB of L2-miss fetch per 64 frames for 7.7 MB
// of distinct lines per frame.  This kernel ranks the keypoints of a (frame, level) by (tile row, tile column) of 128-byte x 32-row tiles,
// list order within a tile (a stable rank: N <= ~450 per level, every thread counts the keys in front of its own in LDS), and writes the
// packed keypoint and its LIST index in visiting order.  The output position of a keypoint stays its list position: results unchanged.
__global__ __launch_bounds__(256) void k_kp_order(const Geom* __restrict__ g, const uint32_t* __restrict__ lvlKp, const int* __restrict__ lvlKpCount,
                                                  uint32_t* __restrict__ sortedKp, uint32_t* __restrict__ sortedIdx) {
  __shared__ uint32_t key[kMaxQuota + 8];
  const int f = blockIdx.x, level = blockIdx.y;
  const LevelGeom& L = g->lv[level];
  const int n = min(lvlKpCount[f * g->nlevels + level], min(L.N + 4, kMaxQuota + 8));
  const uint32_t* in = lvlKp + (uint64_t)f * g->kpBlock + L.kpOff;
  uint32_t* outK = sortedKp + (uint64_t)f * g->kpBlock + L.kpOff;
  uint32_t* outI = sortedIdx + (uint64_t)f * g->kpBlock + L.kpOff;
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t pk = in[i];
    key[i] = ((uint32_t)(pt_y(pk) >> 5) << 20) | ((uint32_t)(pt_x(pk) >> 7) << 12) | (uint32_t)i;   // tile row, tile column, list index (< 4096)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t k = key[i];
    int rank = 0;
    for (int j = 0; j < n; j++) rank += key[j] < k ? 1 : 0;     // keys are distinct (the list index is part of them)
    outK[rank] = in[i];
    outI[rank] = (uint32_t)i;
  }
}

__global__ __launch_bounds__(256) void k_describe(const Geom* __restrict__ g, ImgSrc src, const u8* __restrict__ blur,
                                                  const uint32_t* __restrict__ lvlKp, const uint32_t* __restrict__ lvlKpIdx, const int* __restrict__ lvlKpCount,
                                                  dvs_keypoint* __restrict__ outKp, u8* __restrict__ outDesc,
                                                  int* __restrict__ nOut, int capacity) {
  typedef uint4 __attribute__((aligned(1))) uint4u;
  __shared__ __attribute__((aligned(16))) u8 win[4][2][kWinRows * kWinPitch];
  DVS_CHAIN_PRIO();
  const int wg = xcd_contiguous_id();
  const int f = wg / (int)gridDim.x, bx = wg - f * (int)gridDim.x;
  const int lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nl = g->nlevels;
  const int* cnt = lvlKpCount + f * nl;
  const int slot0 = (bx * 4 + wv) * kDescKP;
  if (bx == 0 && threadIdx.x == 0) {
    if (src.slotted) {
      for (int l = 0; l < nl; l++) nOut[f * nl + l] = cnt[l];   // per-level counts of the slotted block
    } else {
      int total = 0;
      for (int l = 0; l < nl; l++) total += cnt[l];
      nOut[f] = min(total, capacity);
    }
  }
  if (slot0 >= g->kpBlock) return;
  // ---- slot resolve (lane & 7 = keypoint of the wave; lanes >= 8 repeat lanes 0..7)
  const int slot = min(slot0 + (lane & 7), g->kpBlock - 1);
  // (lvlKp / lvlKpIdx: the level's keypoints in VISITING order and their list positions, k_kp_order)
  const uint32_t pkRaw = lvlKp[(uint64_t)f * g->kpBlock + slot];  // requested before the counts it is validated against
  const uint32_t idxRaw = lvlKpIdx[(uint64_t)f * g->kpBlock + slot];
  int level = 0, pre = 0, acc = 0;
  for (int l = 0; l < nl; l++) {
    if (slot >= g->lv[l].kpOff) { level = l; pre = acc; }
    acc += cnt[l];
  }
  const LevelGeom& L = g->lv[level];
  const bool inlist = slot - L.kpOff < cnt[level];
  const int idx = inlist ? (int)idxRaw : slot - L.kpOff;          // list position of the keypoint visited at this slot
  const int gi = src.slotted ? L.kpOff + idx : pre + idx;
  const bool valid = slot0 + (lane & 7) < g->kpBlock && inlist && gi < capacity;
  if ((__ballot(valid) & 0xffull) == 0) return;
  const uint32_t pk = valid ? pkRaw : 0u;
  // an empty slot reads the patch of (level 0, first legal position): in bounds, result discarded
  const int lvl = valid ? level : 0;
  const int x = pt_x(pk) + kMinBorder + (valid ? 0 : 3), y = pt_y(pk) + kMinBorder + (valid ? 0 : 3);  // level pixels (:886-887)
  const uint64_t pyrOff = (uint64_t)f * g->frameBytes + g->lv[lvl].off;
  const u8* imgL = lvl == 0 ? src.img0 + (uint64_t)f * src.fstride0 : src.pyr + pyrOff;
  const int pitchL = lvl == 0 ? (int)src.step0 : g->lv[lvl].pitch;
  const u8* blurL = blur + pyrOff;
  const int bpL = g->lv[lvl].pitch;
#define DVS_RL(v, i) __builtin_amdgcn_readlane((int)(v), (i))
#define DVS_RLP(p, i) reinterpret_cast<const u8*>(((uint64_t)(uint32_t)DVS_RL((uint32_t)((uint64_t)(p) >> 32), i) << 32) | (uint32_t)DVS_RL((uint32_t)(uint64_t)(p), i))
  // ---- orientation patches, all eight requested before any is consumed
  const int prow = min(lane >> 1, 2 * kHalfPatch), half = lane & 1;
  const int v = prow - kHalfPatch;
  const int pcol = half ? 1 : -kHalfPatch;
  uint4 d[kDescKP];
#pragma unroll
  for (int i = 0; i < kDescKP; i++) {
    const u8* ib = DVS_RLP(imgL, i);
    const int pi = DVS_RL(pitchL, i), xi = DVS_RL(x, i), yi = DVS_RL(y, i);
    d[i] = *reinterpret_cast<const uint4u*>(ib + (uint32_t)(mul_i24(yi + v, pi) + xi + pcol));  // all terms >= 0, 24-bit product
  }
  // blurred window of keypoint 0
  const int e0 = lane, e1 = 64 + lane;  // 111 = 37 rows x 3 sixteen-byte pieces
  const int wr0 = e0 / 3, wc0 = e0 - 3 * wr0, wr1 = min(e1 / 3, kWinRows - 1), wc1 = e1 - 3 * (e1 / 3);
  const bool w1ok = e1 < kWinRows * 3;
  static_assert(kDescKP == 8, "the window registers below are named per keypoint");
  uint4 wq0_0, wq0_1, wq0_2, wq0_3, wq0_4, wq0_5, wq0_6, wq0_7, wq1_0, wq1_1, wq1_2, wq1_3, wq1_4, wq1_5, wq1_6, wq1_7;
#define DVS_REQUEST_WINDOW(i)                                                                                  \
  {                                                                                                            \
    const u8* bb = DVS_RLP(blurL, i);                                                                          \
    const int bp = DVS_RL(bpL, i), xi = DVS_RL(x, i), yi = DVS_RL(y, i);                                       \
    const int xa = (xi - kWinR) & ~3; /* dword-aligned window origin; 48 bytes per row cover x-18 .. x+18 */  \
    const u8* o = bb + (int64_t)(yi - kWinR) * bp + xa;                                                        \
    wq0_##i = *reinterpret_cast<const uint4*>(o + (uint32_t)(mul_i24(wr0, bp) + 16 * wc0));                    \
    wq1_##i = *reinterpret_cast<const uint4*>(o + (uint32_t)(mul_i24(wr1, bp) + 16 * wc1));                    \
  }
  // a wave's life is one latency chain (slot -> patches / windows -> samples), so every window is requested up front: the
  // first half behind the patches, the second half into the registers the patches free
  DVS_REQUEST_WINDOW(0) DVS_REQUEST_WINDOW(1) DVS_REQUEST_WINDOW(2) DVS_REQUEST_WINDOW(3)
  // pattern -> floats, once per wave
  float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int pat = reinterpret_cast<const int*>(c_pattern)[64 * r + lane];
    px0[r] = (float)(int8_t)(pat & 0xff); py0[r] = (float)(int8_t)((pat >> 8) & 0xff);
    px1[r] = (float)(int8_t)((pat >> 16) & 0xff); py1[r] = (float)(int8_t)((pat >> 24) & 0xff);
  }
  // ---- IC_Angle (ORBextractor.cpp:76-103): membership and the u weights are per-lane byte tables (Geom::icw):
  // sum u*I = sum (u+15)*I - 15 * sum I, exact integers
  const uint32_t* wt = g->icw[lane];
  const uint32_t w0 = wt[0], w1 = wt[1], w2 = wt[2], w3 = wt[3], w4 = wt[4], w5 = wt[5], w6 = wt[6], w7 = wt[7];
  int q[2 * kDescKP];  // [i] = m10 of keypoint i, [8 + i] = m01
#pragma unroll
  for (int i = 0; i < kDescKP; i++) {
    uint32_t su = __builtin_amdgcn_udot4(d[i].x, w0, 0u, false);
    su = __builtin_amdgcn_udot4(d[i].y, w1, su, false);
    su = __builtin_amdgcn_udot4(d[i].z, w2, su, false);
    su = __builtin_amdgcn_udot4(d[i].w, w3, su, false);
    uint32_t sm = __builtin_amdgcn_udot4(d[i].x, w4, 0u, false);
    sm = __builtin_amdgcn_udot4(d[i].y, w5, sm, false);
    sm = __builtin_amdgcn_udot4(d[i].z, w6, sm, false);
    sm = __builtin_amdgcn_udot4(d[i].w, w7, sm, false);
    q[i] = (int)su - kHalfPatch * (int)sm;
    q[kDescKP + i] = v * (int)sm;
  }
  DVS_REQUEST_WINDOW(4) DVS_REQUEST_WINDOW(5) DVS_REQUEST_WINDOW(6) DVS_REQUEST_WINDOW(7)
  // transposing butterfly: after the step with lane bit B, a lane keeps the half of the quantities selected by its bit B
#define DVS_BFLY(n, o)                                        \
  {                                                           \
    const bool hi = (lane & (o)) != 0;                        \
    _Pragma("unroll") for (int j = 0; j < (n); j++) {         \
      const int keep = hi ? q[(n) + j] : q[j];                \
      const int send = hi ? q[j] : q[(n) + j];                \
      q[j] = keep + __shfl_xor(send, (o));                    \
    }                                                         \
  }
  DVS_BFLY(8, 32) DVS_BFLY(4, 16) DVS_BFLY(2, 8) DVS_BFLY(1, 4)
#undef DVS_BFLY
  int tot = q[0];
  tot += __shfl_xor(tot, 2);
  tot += __shfl_xor(tot, 1);
  // lane bits 5 | 4 3 2 = (m01 ? : m10) | keypoint index bits 2 1 0  (bit 4 chose between i and i+4, bit 3 i and i+2, bit 2 i and i+1)
  const int other = __shfl_xor(tot, 32);
  const int m10 = lane < 32 ? tot : other, m01 = lane < 32 ? other : tot;
  const float angleK = fast_atan2_deg((float)m01, (float)m10);  // keypoint (lane >> 2) & 7
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float arad = __fmul_rn(angleK, factorPI);
  const float cosK = gsc::cosf_(arad), sinK = gsc::sinf_(arad);
  // ---- steered BRIEF on the blurred level (:107-146), one keypoint after the other
  const unsigned vmask = (unsigned)(__ballot(valid) & 0xffull);
  constexpr int kCsLane = 4;  // lane stride of the per-keypoint (cos, sin, angle)
  const int giL = gi;
  auto brief = [&](const int i, const uint4& wa, const uint4& wb) __attribute__((always_inline)) {
    u8* wl = win[wv][i & 1];
    *reinterpret_cast<uint4*>(wl + wr0 * kWinPitch + 16 * wc0) = wa;
    if (w1ok) *reinterpret_cast<uint4*>(wl + wr1 * kWinPitch + 16 * wc1) = wb;
    const int xi = DVS_RL(x, i);
    const int wxi = xi - ((xi - kWinR) & ~3);
    wave_lds_fence();
    const float a = __builtin_bit_cast(float, DVS_RL(__builtin_bit_cast(int, cosK), kCsLane * i));
    const float b = __builtin_bit_cast(float, DVS_RL(__builtin_bit_cast(int, sinK), kCsLane * i));
    // cvRound of the steered coordinates without a conversion: fl(t + (2^23 + 32)) IS 2^23 + 32 + rint(t) (one ulp = 1 there, ties to
    // even, and 2^23 + 32 is even), so the low 24 bits of its encoding are 32 + cvRound(t) > 0 and v_mad_i32_i24 — which reads exactly
    // those bits of its two factors and ALL 32 bits of its addend — forms (32 + row) * pitch + (0x4B000020 + col) from the two sums
    // directly; the constant comes off the base.  (v_rndne + v_cvt per coordinate before: 16 instructions per keypoint and lane.)
    const int bco = kWinR * kWinPitch + wxi - 32 * kWinPitch - 0x4B000020;
    unsigned long long words[4];
    // the two points of a test side by side on the packed-f32 pipe (v_pk_mul_f32 / v_pk_add_f32: two IEEE single operations per
    // instruction, each rounded exactly as the scalar form — the library is built without contraction): 8 instead of 16 vector
    // instructions per round for the same four products, two sums / differences and two rounding adds per coordinate pair
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a2 = {a, a}, b2 = {b, b};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      constexpr float kRound = 8388640.0f;   // 2^23 + 32
      const f32x2 kR2 = {kRound, kRound};
      const f32x2 pxv = {px0[r], px1[r]}, pyv = {py0[r], py1[r]};
      const f32x2 rowv = (pxv * b2 + pyv * a2) + kR2;   // (row of point 0, row of point 1)
      const f32x2 colv = (pxv * a2 - pyv * b2) + kR2;
      const float fr0 = rowv[0], fr1 = rowv[1], fc0 = colv[0], fc1 = colv[1];
      const int r0 = __builtin_bit_cast(int, fr0), r1 = __builtin_bit_cast(int, fr1);
      const int c0 = __builtin_bit_cast(int, fc0), c1 = __builtin_bit_cast(int, fc1);
      const int t0 = wl[mad_i24(r0, kWinPitch, c0) + bco];
      const int t1 = wl[mad_i24(r1, kWinPitch, c1) + bco];
      words[r] = __ballot(t0 < t1);
    }
    if ((vmask >> i) & 1u) {
      const int gii = DVS_RL(giL, i);
      if (lane < 4) {
        const unsigned long long w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
        reinterpret_cast<unsigned long long*>(outDesc + ((uint64_t)f * capacity + gii) * 32)[lane] = w;
      }
    }
  };
  brief(0, wq0_0, wq1_0); brief(1, wq0_1, wq1_1); brief(2, wq0_2, wq1_2); brief(3, wq0_3, wq1_3);
  brief(4, wq0_4, wq1_4); brief(5, wq0_5, wq1_5); brief(6, wq0_6, wq1_6); brief(7, wq0_7, wq1_7);
  // ---- keypoints: lane i < 8 writes keypoint i; its angle lives in lane 4 * i
  const float angle = __shfl(angleK, kCsLane * (lane & 7));
  if (lane < kDescKP && valid) {
    dvs_keypoint kp;
    kp.x = (float)x; kp.y = (float)y;
    if (level != 0) { kp.x = __fmul_rn(kp.x, L.scale); kp.y = __fmul_rn(kp.y, L.scale); }  // pt *= scale (:1148-1150)
    kp.size = L.kpSize;
    kp.angle = angle;
    kp.response = (float)pt_s(pk);
    kp.octave = level;
    kp.class_id = -1;
    outKp[(uint64_t)f * capacity + gi] = kp;
  }
#undef DVS_REQUEST_WINDOW
#undef DVS_RL
#undef DVS_RLP
}

// ---------------------------------------------------------------------------------------------
// level-sharded extraction (SURVEY.md §8e "Partitioning", small batches): every rank extracted its levels into a level-slotted
// block {counts[nimg][nl], keypoints[nimg][kpBlock], descriptors[nimg][kpBlock]}; after the all-gather each rank restores the
// reference's level-major order: thread = (frame, slot) copies its 28 + 32 bytes from the block of the rank that owns the
// slot's level to the compacted position (sum of the lower levels' counts + index)
// ---------------------------------------------------------------------------------------------
struct LevelBlockLayout { uint64_t blockBytes, kpsOff, descOff; int nl, kpBlock; int kpOff[DVS_MAX_LEVELS]; int owner[DVS_MAX_LEVELS]; };

__global__ __launch_bounds__(256) void k_merge_levels(LevelBlockLayout Y, const u8* __restrict__ blocks, int nimg, dvs_keypoint* __restrict__ outKp,
                                                      u8* __restrict__ outDesc, int capacity, int* __restrict__ nOut) {
  const int f = blockIdx.y;
  const int slot = blockIdx.x * 256 + threadIdx.x;
  int level = 0, pre = 0, acc = 0, cntL = 0;
  for (int l = 0; l < Y.nl; l++) {
    const int c = reinterpret_cast<const int*>(blocks + (uint64_t)Y.owner[l] * Y.blockBytes)[f * Y.nl + l];
    if (slot >= Y.kpOff[l]) { level = l; pre = acc; cntL = c; }
    acc += c;
  }
  if (slot == 0) nOut[f] = min(acc, capacity);
  if (slot >= Y.kpBlock) return;
  const int idx = slot - Y.kpOff[level], gi = pre + idx;
  if (idx >= cntL || gi >= capacity) return;
  const u8* blk = blocks + (uint64_t)Y.owner[level] * Y.blockBytes;
  const uint32_t* sk = reinterpret_cast<const uint32_t*>(blk + Y.kpsOff) + ((uint64_t)f * Y.kpBlock + slot) * 7;
  uint32_t* dk = reinterpret_cast<uint32_t*>(outKp) + ((uint64_t)f * capacity + gi) * 7;
#pragma unroll
  for (int k = 0; k < 7; k++) dk[k] = sk[k];
  const uint4* sd = reinterpret_cast<const uint4*>(blk + Y.descOff) + ((uint64_t)f * Y.kpBlock + slot) * 2;
  uint4* dd = reinterpret_cast<uint4*>(outDesc) + ((uint64_t)f * capacity + gi) * 2;
  dd[0] = sd[0]; dd[1] = sd[1];
}


// =============================================================================================
// Host entry points (dvs_orb_extract[_batch]): the frames' results leave through this kernel — it writes the n_out[f] keypoints and
// descriptors of every frame (not the whole capacity) and the counts straight into the handle's pinned host block, and the LAST
// workgroup to finish (ticket) publishes a sequence number behind a system-scope fence, which the host polls.  Replaces three
// device-to-host copy commands and a stream wait per call.
__global__ __launch_bounds__(256) void k_export_host(int nimg, int cap, const uint32_t* __restrict__ kps, const uint32_t* __restrict__ desc,
                                                     const int* __restrict__ nout, uint32_t* __restrict__ hkps, uint32_t* __restrict__ hdesc,
                                                     int* __restrict__ hnout, int* __restrict__ ticket, int* __restrict__ hseq, int seq) {
  const int f = blockIdx.y;
  const int n = min(nout[f], cap);
  const uint32_t* ks = kps + (uint64_t)f * cap * 7;
  const uint32_t* ds = desc + (uint64_t)f * cap * 8;
  uint32_t* kd = hkps + (uint64_t)f * cap * 7;
  uint32_t* dd = hdesc + (uint64_t)f * cap * 8;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * 7; i += gridDim.x * 256) kd[i] = ks[i];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * 8; i += gridDim.x * 256) dd[i] = ds[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) hnout[f] = nout[f];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = gridDim.x * gridDim.y;
    if (atomicAdd(ticket, 1) == total - 1) {
      *ticket = 0;
      __threadfence_system();
      *reinterpret_cast<volatile int*>(hseq) = seq;
      __threadfence_system();
    }
  }
}

}  // namespace dvs
