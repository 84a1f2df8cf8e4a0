// match.hip — brute-force Hamming matcher (boundary B2, include/dvslam_hip.h).
// Replaces cv::BFMatcher(cv::NORM_HAMMING[, false]).match(query, train, matches) at the reference's
// call sites src/frontend.cpp:614, src/frontend.cpp:1123, src/backend.cpp:1072 (ctor frontend.cpp:220,
// backend.cpp:222).  Semantics (OpenCV batchDistance, K = 1): per query the minimum popcount(q XOR t)
// over train rows scanned in increasing index with a strict '<' update, i.e. lowest index on ties.
//
// Kernel shape (k_match<kSplit, kQPL>): workgroup = 64 * kQPL queries x kSplit train slices, one wavefront per slice.
// Each lane keeps kQPL 256-bit queries in VGPRs; the train row index is wave-uniform, so train rows arrive by scalar
// loads (software pipelined) and the inner loop is 8 x (v_xor, v_bcnt accumulate) per query and train row.  The waves
// scan disjoint, ordered slices of the train set and merge through LDS in slice order (keeps the lowest-index
// tie-break exact).  <8, 2> is the batch shape, <16, 1> the single-job (latency) shape.
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include <atomic>
#include <algorithm>
#include "common.h"
#include "io_pinned.h"

namespace dvs {

typedef unsigned long long u64;

// kSplit: train-set slices per workgroup (one wavefront each): finer waves balance the last scheduling round
// kQPL:   queries per lane: every scalar train row feeds kQPL independent popcount chains
// <8, 2> is the throughput shape (batches of jobs); <16, 1> quadruples the wavefronts of a job for the live pattern of ONE
// 2000 x 2000 job, which otherwise occupies an eighth of the SIMDs (0.041 -> see tools/latency.py)
template <int kSplit, int kQPL>
__global__ __launch_bounds__(64 * kSplit) void k_match(const u64* __restrict__ q, const int* __restrict__ nqArr, int nqConst, int qStrideRows,
                                                       const u64* __restrict__ t, const int* __restrict__ ntArr, int ntConst, int tStrideRows,
                                                       int* __restrict__ outIdx, int* __restrict__ outDist,
                                                       const u64* __restrict__ t0 = nullptr, const int* __restrict__ nt0 = nullptr) {
  __shared__ int sd[kSplit][64 * kQPL];
  __shared__ int si[kSplit][64 * kQPL];
  const int pair = blockIdx.y;
  const int nq = nqArr ? min(nqArr[pair], qStrideRows) : nqConst;
  // job 0's train set may live elsewhere (frame sequences: the previous batch's last frame), t0 / nt0 then replace t / ntArr
  const bool first = pair == 0 && t0 != nullptr;
  // counts read from device memory are clamped to the row strides the caller declared (a corrupt count must not walk out of the
  // buffers); the predecessor block of a sequence has no declared size and is trusted
  const int nt = first ? *nt0 : (ntArr ? min(ntArr[pair], tStrideRows > 0 ? tStrideRows : ntArr[pair]) : ntConst);
  const int q0 = blockIdx.x * 64 * kQPL;
  if (q0 >= nq) return;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index as an SGPR: train addresses become scalar
  u64 a[kQPL][4];
#pragma unroll
  for (int u = 0; u < kQPL; u++) {
    const int qi = q0 + u * 64 + lane;
    const u64* qp = q + ((size_t)pair * qStrideRows + (qi < nq ? qi : q0)) * 4;
    a[u][0] = qp[0]; a[u][1] = qp[1]; a[u][2] = qp[2]; a[u][3] = qp[3];
  }
  const int chunk = (nt + kSplit - 1) / kSplit;
  const int jb = min(nt, w * chunk), je = min(nt, jb + chunk);
  const u64* tp = first ? t0 : t + (ptrdiff_t)pair * tStrideRows * 4;
  // running best as ONE word: distance << 23 | train index (nt < 2^23, checked on the host) -> a single v_min_u32 per pair
  // keeps the smallest distance and, on ties, the lowest index; the eight popcounts chain through v_bcnt's accumulator.
  unsigned bestp[kQPL];
#pragma unroll
  for (int u = 0; u < kQPL; u++) bestp[u] = 0xFFFFFFFFu;
  auto dist = [&](int u, const u64* r) -> unsigned {
    unsigned d;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 x = a[u][k] ^ r[k];
      // v_bcnt_u32_b32 d, x, d : popcount with accumulate (hipcc otherwise emits separate v_add3 trees); the first one adds to
      // the inline constant 0 instead of a zeroed register
      if (k == 0) asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"((unsigned)x));
      else asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"((unsigned)x));
      asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"((unsigned)(x >> 32)));
    }
    return d;
  };
  int j = jb;
  if (j + 4 <= je) {  // software pipeline: the scalar loads of trip i+1 are issued before the popcounts of trip i
    u64 rr[16], nx[16];
    {
      const u64* r = tp + (size_t)j * 4;
#pragma unroll
      for (int k = 0; k < 16; k++) rr[k] = r[k];
    }
    for (; j + 4 <= je; j += 4) {
      const bool more = j + 8 <= je;
      const u64* r = tp + (size_t)(more ? j + 4 : j) * 4;
#pragma unroll
      for (int k = 0; k < 16; k++) nx[k] = r[k];
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int u = 0; u < kQPL; u++) bestp[u] = min(bestp[u], (dist(u, &rr[4 * k]) << 23) | (unsigned)(j + k));
#pragma unroll
      for (int k = 0; k < 16; k++) rr[k] = nx[k];
    }
  }
  for (; j < je; j++) {
    const u64* r = tp + (size_t)j * 4;
#pragma unroll
    for (int u = 0; u < kQPL; u++) bestp[u] = min(bestp[u], (dist(u, r) << 23) | (unsigned)j);
  }
  int best[kQPL], bi[kQPL];
#pragma unroll
  for (int u = 0; u < kQPL; u++) {
    best[u] = bestp[u] == 0xFFFFFFFFu ? INT_MAX : (int)(bestp[u] >> 23);
    bi[u] = bestp[u] == 0xFFFFFFFFu ? -1 : (int)(bestp[u] & 0x7FFFFFu);
  }
#pragma unroll
  for (int u = 0; u < kQPL; u++) { sd[w][u * 64 + lane] = best[u]; si[w][u * 64 + lane] = bi[u]; }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int u = 0; u < kQPL; u++) {
      const int qi = q0 + u * 64 + lane;
      if (qi >= nq) continue;
      int bd = best[u], bidx = bi[u];
#pragma unroll
      for (int k = 1; k < kSplit; k++) {  // slices are ordered by train index: strict '<' keeps the lowest index on ties
        const int d = sd[k][u * 64 + lane];
        if (d < bd) { bd = d; bidx = si[k][u * 64 + lane]; }
      }
      outIdx[(size_t)pair * qStrideRows + qi] = bidx;
      outDist[(size_t)pair * qStrideRows + qi] = bd;
    }
  }
}

// The FEW-jobs match with the train set staged in LDS and 16 queries per workgroup.  k_match<16, 1> gives one 2000 x 2000 job 32
// workgroups of 16 wavefronts — 32 of 256 CUs, each wavefront walking 125 train rows through scalar loads one trip ahead: 18.5 us per
// launch whatever the job count up to 8 (tools/time_match_few.py).  Here a workgroup takes 16 queries: its 1024 threads copy the job's
// train set (<= ldsRows x 32 bytes: one coalesced round trip) into LDS, lane = (query, quarter): the 64 (wavefront, quarter) pairs each
// scan 1/64 of the rows for their query, the four quarters fold by two shuffles, the sixteen wavefronts through LDS.  The running best
// is ONE word — distance << 23 | train index — so every fold is an unsigned minimum: smallest distance, lowest index on ties, in any order.
// 125 workgroups per job.  A job whose train set does not fit (nt > ldsRows: the trusted predecessor block of a sequence) reads global memory.
constexpr int kLdsQ = 16;   // queries per workgroup
__global__ __launch_bounds__(1024) void k_match_lds(const u64* __restrict__ q, const int* __restrict__ nqArr, int nqConst, int qStrideRows,
                                                    const u64* __restrict__ t, const int* __restrict__ ntArr, int ntConst, int tStrideRows,
                                                    int* __restrict__ outIdx, int* __restrict__ outDist, int ldsRows,
                                                    const u64* __restrict__ t0 = nullptr, const int* __restrict__ nt0 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) u64 trows[];
  __shared__ unsigned sp[16][kLdsQ];
  const int pair = blockIdx.y;
  const int nq = nqArr ? min(nqArr[pair], qStrideRows) : nqConst;
  const bool first = pair == 0 && t0 != nullptr;
  const int nt = first ? *nt0 : (ntArr ? min(ntArr[pair], tStrideRows > 0 ? tStrideRows : ntArr[pair]) : ntConst);
  const int q0 = blockIdx.x * kLdsQ;
  if (q0 >= nq) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ql = lane & (kLdsQ - 1), quarter = lane >> 4;
  const u64* tp = first ? t0 : t + (ptrdiff_t)pair * tStrideRows * 4;
  const bool staged = nt <= ldsRows;
  if (staged) {   // 16 bytes per thread and trip (descriptor rows are 32-byte aligned)
    const uint4* src = reinterpret_cast<const uint4*>(tp);
    uint4* dst = reinterpret_cast<uint4*>(trows);
    for (int i = threadIdx.x; i < 2 * nt; i += 1024) dst[i] = src[i];
  }
  const int qi = q0 + ql;
  const u64* qp = q + ((size_t)pair * qStrideRows + (qi < nq ? qi : q0)) * 4;
  const u64 a0 = qp[0], a1 = qp[1], a2 = qp[2], a3 = qp[3];
  __syncthreads();
  const int slice = w * 4 + quarter;            // 64 slices of the train set
  const int chunk = (nt + 63) / 64;
  const int jb = min(nt, slice * chunk), je = min(nt, jb + chunk);
  unsigned bestp = 0xFFFFFFFFu;
  auto dist = [&](u64 r0, u64 r1, u64 r2, u64 r3) -> unsigned {
    return (unsigned)(__popcll(a0 ^ r0) + __popcll(a1 ^ r1) + __popcll(a2 ^ r2) + __popcll(a3 ^ r3));
  };
  const u64* rows = staged ? trows : tp;
  if (staged) {
    int j = jb;
    for (; j + 4 <= je; j += 4) {   // four rows' reads in flight
      const ulonglong2* r = reinterpret_cast<const ulonglong2*>(trows + (size_t)j * 4);
      ulonglong2 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = r[k];
#pragma unroll
      for (int k = 0; k < 4; k++) bestp = min(bestp, (dist(v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y) << 23) | (unsigned)(j + k));
    }
    for (; j < je; j++) {
      const u64* r = trows + (size_t)j * 4;
      bestp = min(bestp, (dist(r[0], r[1], r[2], r[3]) << 23) | (unsigned)j);
    }
  } else {
    for (int j = jb; j < je; j++) {
      const u64* r = rows + (size_t)j * 4;
      bestp = min(bestp, (dist(r[0], r[1], r[2], r[3]) << 23) | (unsigned)j);
    }
  }
  bestp = min(bestp, (unsigned)__shfl_xor((int)bestp, 16));
  bestp = min(bestp, (unsigned)__shfl_xor((int)bestp, 32));
  if (quarter == 0) sp[w][ql] = bestp;
  __syncthreads();
  if (w == 0 && quarter == 0 && qi < nq) {
    unsigned b = sp[0][ql];
#pragma unroll
    for (int k = 1; k < 16; k++) b = min(b, sp[k][ql]);
    outIdx[(size_t)pair * qStrideRows + qi] = b == 0xFFFFFFFFu ? -1 : (int)(b & 0x7FFFFFu);
    outDist[(size_t)pair * qStrideRows + qi] = b == 0xFFFFFFFFu ? INT_MAX : (int)(b >> 23);
  }
}

// enqueue the few-jobs match: train sets of up to 4096 rows are staged in LDS (k_match_lds), larger strides take k_match<16, 1>
static inline void launch_match_few(hipStream_t st, dim3 grid, const u64* q, const int* nqArr, int nqConst, int qStrideRows, const u64* t,
                                    const int* ntArr, int ntConst, int tStrideRows, int* outIdx, int* outDist, int trainRows,
                                    const u64* t0 = nullptr, const int* nt0 = nullptr) {
  const bool lds_on = dvs::env_switch("DVS_MATCH_LDS", 1) != 0;
  // up to 6 jobs (what a lane of dvs_pipeline enqueues, and the single-call entry point): one 2000 x 2000 job 18.3 -> 7.7 us per launch, two
  // 18.5 -> 8.1, four 18.8 -> 13.1, six 18.7 -> 19.5 alone but +8 % in the six-frame lane step (no scalar-load chain beside the other lanes'
  // kernels); 7 and 8 jobs (the four-stream form's match stream) lose 3-7 %: k_match<16, 1> there.  DVS_MATCH_LDS=0: never.
  // the train set + the kernel's own 1 KB must fit the device's LDS per workgroup; the limit and the kernel's attribute are per DEVICE
  // (a function attribute set on one device says nothing about the next): looked up once for each
  static std::atomic<int> ldsLimit[64];   // 0 = not asked yet, -1 = k_match_lds unusable there
  int dev = 0;
  (void)hipGetDevice(&dev);
  int lim = dev >= 0 && dev < 64 ? ldsLimit[dev].load(std::memory_order_acquire) : -1;
  if (lim == 0) {
    int v = 0;
    lim = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 1024 ? v : -1;
    if (lim > 0 && hipFuncSetAttribute((const void*)k_match_lds, hipFuncAttributeMaxDynamicSharedMemorySize, std::min(lim - 1024, 4096 * 32)) != hipSuccess) lim = -1;
    (void)hipGetLastError();
    ldsLimit[dev].store(lim, std::memory_order_release);
  }
  if (lds_on && grid.y <= 6 && trainRows > 0 && trainRows <= 4096 && (long long)trainRows * 32 + 1024 <= lim && (((uintptr_t)t | (uintptr_t)t0) & 15) == 0) {
    const dim3 g16((grid.x * 64 + kLdsQ - 1) / kLdsQ, grid.y);   // (the callers' grids count 64 queries per workgroup)
    hipLaunchKernelGGL(k_match_lds, g16, dim3(1024), (size_t)trainRows * 32, st, q, nqArr, nqConst, qStrideRows, t, ntArr, ntConst, tStrideRows, outIdx,
                       outDist, trainRows, t0, nt0);
  } else {
    hipLaunchKernelGGL((k_match<16, 1>), grid, dim3(1024), 0, st, q, nqArr, nqConst, qStrideRows, t, ntArr, ntConst, tStrideRows, outIdx, outDist, t0, nt0);
  }
}

// =============================================================================================================================
// Matrix-core match for batches of large jobs (rebuilt twice in round 5).  north_star says "no MFMA: none of this is
// a dense contraction" — the match is one once the bits are matrix elements: a 2000 x 256 . 256 x 2000 product with exact accumulation
// replaces 16 M (v_xor, v_bcnt) pairs per job on the vector pipe, which every other kernel of the path saturates while the matrix cores
// sit idle.  DVS_MATCH_MFMA=0 selects the popcount kernels.
//   dist(q, t) = |q| + |t| - 2 |q AND t|,    key = 128 |q AND t| - 64 |t| + 63 - code(row)  [- 32768 for rows past the count],
// code = the row's place among the 64 rows a lane's accumulators hold per 128-row chunk: ONE maximum over the accumulator registers picks
// the smallest distance and, among equals, the lowest row (cv::BFMatcher's tie-break); dist = |q| - (key >> 6).  Rows past the count
// repeat row nt - 1 (clamped loads): a repeat can never beat its original.
// Rounds 2-4: int8 MFMA on +8 / -8 byte images that a second kernel wrote to HBM (41.6 MB written + 76 MB read per 64 jobs for 9.2 MB of
// descriptors), streamed through 72 KB of LDS per workgroup — 57 us (12 %) of the 64-frame step beside FAST.  Round 5, first step: int8
// operands built in the kernel, 18 KB of LDS, one launch (-5.3 % step).  Second step, this kernel: the matrix pipe's TIME shows in the
// step — doubling the int8 kernel's matrix work (2.3 M instructions of 17.4 ns = 61 us of every SIMD's matrix pipe per 64 jobs)
// lengthened the step by 37 us (profiles/r05_mfma_exposure_ab.log) — and v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands
// contracts 64 k-values in the time the int8 form takes for 32 (tools/probe/mfma_f8_probe.hip: 16.4 ns; semantics, operand and result
// layouts checked there against a host sum).  A descriptor BIT is one FP4 nibble: set -> 0x2 (1.0), clear -> 0; the query operand's
// block scale is 2^7, so four k-steps give 128 |q AND t| as exact f32 integers (everything stays below 2^24).  The per-row terms of the
// key no longer take a k-step of their own: they are the C operand — the wavefront whose turn it is computes the tile's 32 row constants
// (it holds the 16 bytes of every row half: popcount + one cross-half swap), all four read them as the accumulators' initial value.
// Contraction order: lane (h, r) of a fragment supplies 32 k-values of row r; which bits they are is free as long as both operands
// agree, so half h takes bytes [16 h, 16 h + 16) of the row and k-step j its word j.
// Per 32-row tile and wavefront: ONE ds_write_b128 (word w of every row half -> 32 nibbles through a 256-entry LUT in LDS: 8 bits ->
// 8 nibbles), 4 + 4 ds_read_b128, 4 NQ matrix instructions of 16.4 ns (int8: 9 NQ of 17.4), 8 NQ v_max3_f32; two 4.1-KB tile buffers,
// one barrier per tile.  The query fragments are 4 registers per k-step, so NQ = 4 tiles fit a wavefront without spills (248 VGPRs): the
// kernel of >= 32 jobs; fewer jobs take NQ = 2 (twice the workgroups: at 8..31 jobs the machine is not full and latency counts).
// Same-box (64-frame step): 0.4673 ms (rounds 2-4's kernels) -> 0.4426 (int8 in-kernel) -> 0.436 (this).  HBM: 9.2 MB per 64 jobs = the
// algorithmic bytes.  Bit-exact against the oracle: tests/test_gpu_match.py, tests/test_gpu_pipeline.py.
// =============================================================================================================================
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NQ>
__global__ __launch_bounds__(256) void k_match_fp4(const uint8_t* __restrict__ q, const int* __restrict__ nqArr, int qStrideRows,
                                                   const uint8_t* __restrict__ t, const int* __restrict__ ntArr, int tStrideRows,
                                                   const uint8_t* __restrict__ t0, const int* __restrict__ nt0,
                                                   int* __restrict__ outIdx, int* __restrict__ outDist) {
  constexpr int kTile = 4 * 1024 + 128;   // four 1-KB fragments (k-steps) + the 32 row constants (f32) of a 32-row tile
  __shared__ uint32_t lut[256];           // 8 descriptor bits -> 8 FP4 nibbles of 0 / 1.0
  __shared__ __attribute__((aligned(16))) uint8_t tl[2][kTile];
  int pair = blockIdx.y, qt = blockIdx.x;
  if ((gridDim.y & 7) == 0) {   // pair p entirely on XCD p % 8: its workgroups stream the same train set through ONE L2
    const int lid = blockIdx.x + gridDim.x * blockIdx.y, xcd = lid & 7, k = lid >> 3;
    pair = xcd + 8 * (k / (int)gridDim.x);
    qt = k % (int)gridDim.x;
  }
  const bool first = pair == 0 && t0 != nullptr;
  const int nq = min(max(nqArr[pair], 0), qStrideRows);
  const int nt = min(max(first ? *nt0 : ntArr[pair], 0), tStrideRows);
  if (qt * 128 * NQ >= nq) return;   // (uniform over the workgroup: no barrier is skipped by a part of it)
  const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= ((threadIdx.x >> i) & 1u) ? 2u << (4 * i) : 0u;
    lut[threadIdx.x] = o;
  }
  __syncthreads();
  auto nib = [&](uint32_t x) -> v4i {   // 32 bits -> 32 nibbles
    return v4i{(int)lut[x & 255u], (int)lut[(x >> 8) & 255u], (int)lut[(x >> 16) & 255u], (int)lut[x >> 24]};
  };
  auto word = [](const uint4& v, int i) -> uint32_t { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); };
  const uint8_t* qb = q + (size_t)pair * qStrideRows * 32 + 16 * h;
  const uint8_t* tb = (first ? t0 : t + (ptrdiff_t)pair * tStrideRows * 32) + 16 * h;
  v4i bq[NQ][4];
  int popq[NQ];
  const int qtile0 = (qt * 4 + w) * NQ;
#pragma unroll
  for (int u = 0; u < NQ; u++) {
    const int row = min((qtile0 + u) * 32 + r, nq - 1);   // rows past the count repeat the last one (never written out)
    const uint4 v = *reinterpret_cast<const uint4*>(qb + (size_t)row * 32);
    const int ph = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    popq[u] = ph + __shfl_xor(ph, 32);
#pragma unroll
    for (int j = 0; j < 4; j++) bq[u][j] = nib(word(v, j));
  }
  const int nchunks = (nt + 127) >> 7, ntiles = nchunks * 4;
  int bestc[NQ], besti[NQ];
#pragma unroll
  for (int u = 0; u < NQ; u++) { bestc[u] = INT_MIN; besti[u] = -1; }
  const int lanePart = (r >> 3) * 4 + (r & 3);   // the row's place among the 16 accumulator registers of the lane that holds it
  // this wavefront's k-step of a tile (word w of every row half) and, for the wavefront whose turn it is, the tile's row constants
  auto expand = [&](int tile, const uint4& v, uint8_t* buf) {
    *reinterpret_cast<v4i*>(buf + w * 1024 + lane * 16) = nib(w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w)));
    if ((tile & 3) == w) {
      const int ph = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
      const int pop = ph + __shfl_xor(ph, 32);
      const int code = (tile & 3) * 16 + lanePart;
      const int c = 63 - code - 64 * pop - (tile * 32 + r < nt ? 0 : 32768);
      // row r lives in the accumulators of the lanes of half (r >> 2) & 1, register lanePart
      if (h == 0) *reinterpret_cast<float*>(buf + 4096 + (((r >> 2) & 1) * 16 + lanePart) * 4) = (float)c;
    }
  };
  auto load = [&](int tile) -> uint4 { return *reinterpret_cast<const uint4*>(tb + (size_t)min(tile * 32 + r, nt - 1) * 32); };
  uint4 wnext = make_uint4(0u, 0u, 0u, 0u);
  if (ntiles > 0) {
    expand(0, load(0), tl[0]);
    wnext = load(1);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; c++) {
    float key[NQ];
#pragma unroll
    for (int u = 0; u < NQ; u++) key[u] = -3.0e38f;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int tile = c * 4 + m;
      const uint4 wcur = wnext;
      wnext = load(tile + 2);                                     // two tiles ahead (clamped: always a valid address)
      if (tile + 1 < ntiles) expand(tile + 1, wcur, tl[(tile + 1) & 1]);
      const uint8_t* buf = tl[tile & 1];
      v16f rc;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float4 f = *reinterpret_cast<const float4*>(buf + 4096 + h * 64 + i * 16);
        rc[4 * i] = f.x; rc[4 * i + 1] = f.y; rc[4 * i + 2] = f.z; rc[4 * i + 3] = f.w;
      }
      v16f acc[NQ];
#pragma unroll
      for (int u = 0; u < NQ; u++) acc[u] = rc;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const v4i a4 = *reinterpret_cast<const v4i*>(buf + j * 1024 + lane * 16);
        const v8i a = v8i{a4.x, a4.y, a4.z, a4.w, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < NQ; u++) {
          const v8i b = v8i{bq[u][j].x, bq[u][j].y, bq[u][j].z, bq[u][j].w, 0, 0, 0, 0};
          acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[u], 4, 4, 0, 0x7f7f7f7f, 0, (int)0x86868686u);   // FP4 x FP4, query block scale 2^7
        }
      }
#pragma unroll
      for (int u = 0; u < NQ; u++)
#pragma unroll
        for (int i = 0; i < 16; i++) key[u] = fmaxf(key[u], acc[u][i]);
      __syncthreads();   // tile + 1 is complete for everyone; everyone has read tile's buffer, which tile + 2 overwrites
    }
#pragma unroll
    for (int u = 0; u < NQ; u++) {
      const int k = (int)key[u];   // exact: every term is an integer below 2^24
      const int cval = k >> 6, code = 63 - (k & 63);
      const int row = c * 128 + (code >> 4) * 32 + ((code >> 2) & 3) * 8 + h * 4 + (code & 3);
      if (cval > bestc[u]) { bestc[u] = cval; besti[u] = row; }   // later chunks hold higher rows: strict '>' keeps the lowest on ties
    }
  }
#pragma unroll
  for (int u = 0; u < NQ; u++) {
    const int oc = __shfl_xor(bestc[u], 32), oi = __shfl_xor(besti[u], 32);   // the two lane halves hold interleaved rows of the same query
    int bc = bestc[u], bi = besti[u];
    if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    const int qi = (qtile0 + u) * 32 + r;
    if (lane < 32 && qi < nq) {
      outIdx[(size_t)pair * qStrideRows + qi] = nt > 0 ? bi : -1;
      outDist[(size_t)pair * qStrideRows + qi] = nt > 0 ? popq[u] - bc : INT_MAX;
    }
  }
}

// every (query, train) pair with distance < maxDist (the backend's association loop, backend.cpp:1068-1077, as ONE job), in two passes.
// One WAVEFRONT per query: its lanes take the train descriptors in turn — 64 consecutive 32-byte rows per trip, coalesced — so a map
// of 13 000 landmarks is 204 trips instead of 13 000 iterations of one thread (round 2's form: four workgroups, 1.8 ms per keyframe
// at that map size — half of the replay's GPU time; now a few microseconds).
// pass 1: matches per query
__global__ __launch_bounds__(256) void k_thresh_count(const u64* __restrict__ q, int nq, const u64* __restrict__ t, int nt,
                                                      int maxDist, int* __restrict__ counts) {
  const int qi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (qi >= nq) return;
  const int lane = (int)(threadIdx.x & 63);
  const u64* qp = q + (size_t)qi * 4;
  const u64 a0 = qp[0], a1 = qp[1], a2 = qp[2], a3 = qp[3];
  int c = 0;
  for (int j = lane; j < nt; j += 64) {
    const u64* r = t + (size_t)j * 4;
    const int d = __popcll(a0 ^ r[0]) + __popcll(a1 ^ r[1]) + __popcll(a2 ^ r[2]) + __popcll(a3 ^ r[3]);
    c += d < maxDist ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) counts[qi] = c;
}
// pass 2: write (q, t, dist) triplets at the query's exclusive offset (offsets computed between the passes), train index ascending:
// a trip's hits are ranked by the ballot
__global__ __launch_bounds__(256) void k_thresh_write(const u64* __restrict__ q, int nq, const u64* __restrict__ t, int nt,
                                                      int maxDist, const long long* __restrict__ offs, int* __restrict__ pairs, long long cap) {
  const int qi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (qi >= nq) return;
  const int lane = (int)(threadIdx.x & 63);
  const u64* qp = q + (size_t)qi * 4;
  const u64 a0 = qp[0], a1 = qp[1], a2 = qp[2], a3 = qp[3];
  long long o = offs[qi];
  for (int j0 = 0; j0 < nt; j0 += 64) {
    const int j = j0 + lane;
    int d = 0;
    bool hit = false;
    if (j < nt) {
      const u64* r = t + (size_t)j * 4;
      d = __popcll(a0 ^ r[0]) + __popcll(a1 ^ r[1]) + __popcll(a2 ^ r[2]) + __popcll(a3 ^ r[3]);
      hit = d < maxDist;
    }
    const unsigned long long b = __ballot(hit);
    if (hit) {
      const long long pos = o + (long long)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
      if (pos < cap) { pairs[3 * pos] = qi; pairs[3 * pos + 1] = j; pairs[3 * pos + 2] = d; }
    }
    o += __popcll(b);
  }
}
// single-block exclusive scan of int counts into 64-bit offsets (+ total at offs[n])
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ counts, int n, long long* __restrict__ offs) {
  __shared__ long long part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = tid * per, e = min(n, b + per);
  long long s = 0;
  for (int i = b; i < e; i++) s += counts[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    long long run = 0;
    for (int i = 0; i < 1024; i++) { long long v = part[i]; part[i] = run; run += v; }
    offs[n] = run;
  }
  __syncthreads();
  long long run = part[tid];
  for (int i = b; i < e; i++) { offs[i] = run; run += counts[i]; }
}

}  // namespace dvs

using namespace dvs;

// NQ = 4 query tiles per wavefront from 32 jobs on (a full machine: 64 jobs x 4 workgroups of 512 queries), NQ = 2 below (twice the
// workgroups: 8 frames per step 82 -> 89 k frames/s against NQ = 4, 16 frames 116 -> 119 k; NQ = 1 gains nothing more)
static inline void launch_match_fp4(hipStream_t st, int npairs, const uint8_t* q, const int* nq, int qStrideRows, const uint8_t* t, const int* nt, int tStrideRows,
                                    const uint8_t* t0, const int* nt0, int* idx, int* dist) {
  if (npairs >= 32)
    hipLaunchKernelGGL(dvs::k_match_fp4<4>, dim3((qStrideRows + 511) / 512, npairs), dim3(256), 0, st, q, nq, qStrideRows, t, nt, tStrideRows, t0, nt0, idx, dist);
  else
    hipLaunchKernelGGL(dvs::k_match_fp4<2>, dim3((qStrideRows + 255) / 256, npairs), dim3(256), 0, st, q, nq, qStrideRows, t, nt, tStrideRows, t0, nt0, idx, dist);
}

struct dvs_matcher {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // grow-only staging for the host entry points
  void *d_q = nullptr, *d_t = nullptr, *d_idx = nullptr, *d_dist = nullptr, *d_counts = nullptr, *d_offs = nullptr, *d_pairs = nullptr;
  size_t cq = 0, ct = 0, cidx = 0, ccounts = 0, cpairs = 0;
  void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};  // grow-only buffers of the glue entry points (frontend.hip)
  size_t cscratch[4] = {0, 0, 0, 0};
  void* d_zero = nullptr;  // 64 zero bytes: the empty predecessor of dvs_match_hamming_sequence_device
  int use_mfma = 1;        // DVS_MATCH_MFMA=0 keeps every job on the popcount kernel
  // pinned in / out block of the small host entry points (RANSAC stages): inputs are placed here and imported by a kernel, results
  // are exported by a kernel that publishes a sequence number the host polls — no copy commands, no stream wait
  void* h_io = nullptr; size_t cio = 0;
  int* h_seq = nullptr; int io_seq = 0;
};

namespace {
dvs_status grow(void** p, size_t* cap, size_t need) {
  if (need <= *cap && *p) return DVS_OK;
  if (*p) DVS_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  DVS_HIP(hipMalloc(p, need ? need : 1));
  *cap = need;
  return DVS_OK;
}
}  // namespace

namespace dvs {
dvs_status matcher_scratch(dvs_matcher* m, int slot, size_t bytes, void** out) {
  DVS_TRY(grow(&m->scratch[slot], &m->cscratch[slot], bytes));
  *out = m->scratch[slot];
  return DVS_OK;
}
hipStream_t matcher_stream(dvs_matcher* m) { return m->stream; }
dvs_status matcher_pinned(dvs_matcher* m, size_t bytes, void** out, int** h_seq, int** counter) {
  if (bytes > m->cio || !m->h_io) {
    if (m->h_io) { DVS_HIP(hipStreamSynchronize(m->stream)); DVS_HIP(hipHostFree(m->h_io)); m->h_io = nullptr; m->cio = 0; }
    const size_t cap = std::max<size_t>(bytes + bytes / 2, 65536);
    DVS_HIP(hipHostMalloc(&m->h_io, cap));
    m->cio = cap;
  }
  if (!m->h_seq) { DVS_HIP(hipHostMalloc((void**)&m->h_seq, 64)); *m->h_seq = 0; m->io_seq = 0; }
  *out = m->h_io; *h_seq = m->h_seq; *counter = &m->io_seq;
  return DVS_OK;
}
int matcher_device(dvs_matcher* m) { return m->device; }

// every (query, train) pair with distance < max_dist as (q, t, dist) triplets in (q, t) order, LEFT ON THE DEVICE:
// offsets[nq + 1] (exclusive, 64-bit) and the triplet array.  Host inputs are staged; synchronises once for the total.
dvs_status matcher_thresh_device(dvs_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int max_dist, const long long** d_offs,
                                 const int** d_pairs, long long* total) {
  DVS_HIP(hipSetDevice(m->device));
  DVS_TRY(grow(&m->d_q, &m->cq, (size_t)nq * 32));
  DVS_TRY(grow(&m->d_t, &m->ct, (size_t)nt * 32));
  DVS_TRY(grow(&m->d_counts, &m->ccounts, (size_t)nq * 4));
  if (m->d_offs) { DVS_HIP(hipFree(m->d_offs)); m->d_offs = nullptr; }
  DVS_HIP(hipMalloc(&m->d_offs, ((size_t)nq + 1) * 8));
  DVS_HIP(hipMemcpyAsync(m->d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
  DVS_HIP(hipMemcpyAsync(m->d_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, m->stream));
  const dim3 grid((nq + 3) / 4);   // one wavefront per query
  hipLaunchKernelGGL(k_thresh_count, grid, dim3(256), 0, m->stream, (const u64*)m->d_q, nq, (const u64*)m->d_t, nt, max_dist, (int*)m->d_counts);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, m->stream, (const int*)m->d_counts, nq, (long long*)m->d_offs);
  DVS_HIP(hipGetLastError());
  long long tot = 0;
  DVS_HIP(hipMemcpyAsync(&tot, (long long*)m->d_offs + nq, 8, hipMemcpyDeviceToHost, m->stream));
  DVS_HIP(hipStreamSynchronize(m->stream));
  DVS_TRY(grow(&m->d_pairs, &m->cpairs, (size_t)std::max<long long>(tot, 1) * 12));
  if (tot) {
    hipLaunchKernelGGL(k_thresh_write, grid, dim3(256), 0, m->stream, (const u64*)m->d_q, nq, (const u64*)m->d_t, nt, max_dist,
                       (const long long*)m->d_offs, (int*)m->d_pairs, tot);
    DVS_HIP(hipGetLastError());
    DVS_HIP(hipStreamSynchronize(m->stream));
  }
  *d_offs = (const long long*)m->d_offs; *d_pairs = (const int*)m->d_pairs; *total = tot;
  return DVS_OK;
}
}  // namespace dvs

extern "C" {

dvs_status dvs_matcher_create(int32_t device, dvs_matcher** out) {
  DVS_ARG(out);
  *out = nullptr;
  DVS_TRY(check_device(device));
  dvs_matcher* m = new (std::nothrow) dvs_matcher();
  if (!m) { set_error("out of host memory"); return DVS_ERR_HIP; }
  m->device = device;
  hipError_t e = hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete m; set_error("hipStreamCreate: %s", hipGetErrorString(e)); return DVS_ERR_HIP; }
  m->stream = m->own_stream;
  m->use_mfma = dvs::env_switch("DVS_MATCH_MFMA", 1) != 0;
  *out = m;
  return DVS_OK;
}

dvs_status dvs_matcher_create_on_stream(int32_t device, void* hip_stream, dvs_matcher** out) {
  DVS_ARG(out);
  *out = nullptr;
  DVS_TRY(check_device(device));
  dvs_matcher* m = new (std::nothrow) dvs_matcher();
  if (!m) { set_error("out of host memory"); return DVS_ERR_HIP; }
  m->device = device;
  m->stream = (hipStream_t)hip_stream;  // no stream of its own: every HIP stream is a hardware queue
  m->use_mfma = dvs::env_switch("DVS_MATCH_MFMA", 1) != 0;
  *out = m;
  return DVS_OK;
}

void dvs_matcher_destroy(dvs_matcher* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipStreamSynchronize(m->stream);
  void* ptrs[] = {m->d_q, m->d_t, m->d_idx, m->d_dist, m->d_counts, m->d_offs, m->d_pairs, m->scratch[0], m->scratch[1], m->scratch[2], m->scratch[3], m->d_zero};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (m->h_io) (void)hipHostFree(m->h_io);
  if (m->h_seq) (void)hipHostFree(m->h_seq);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  delete m;
}

dvs_status dvs_matcher_set_stream(dvs_matcher* m, void* s) {
  DVS_ARG(m);
  DVS_HIP(hipSetDevice(m->device));
  DVS_HIP(hipStreamSynchronize(m->stream));
  m->stream = (hipStream_t)s;  // NULL = HIP's legacy default stream
  return DVS_OK;
}
DVS_HOOK dvs_status dvs_matcher_use_own_stream(dvs_matcher* m) {
  DVS_ARG(m);
  DVS_HIP(hipSetDevice(m->device));
  DVS_HIP(hipStreamSynchronize(m->stream));
  if (!m->own_stream) DVS_HIP(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
  m->stream = m->own_stream;
  return DVS_OK;
}
dvs_status dvs_matcher_synchronize(dvs_matcher* m) {
  DVS_ARG(m);
  DVS_HIP(hipSetDevice(m->device));
  DVS_HIP(hipStreamSynchronize(m->stream));
  return DVS_OK;
}

dvs_status dvs_match_hamming_batch_device(dvs_matcher* m, const uint8_t* d_q, const int32_t* d_nq, int32_t q_stride_rows,
                                          const uint8_t* d_t, const int32_t* d_nt, int32_t t_stride_rows, int32_t npairs,
                                          int32_t* d_idx, int32_t* d_dist) {
  DVS_ARG(m && d_q && d_t && d_nq && d_nt && d_idx && d_dist && npairs >= 0 && q_stride_rows > 0 && t_stride_rows >= 0);
  DVS_ARG(t_stride_rows < (1 << 23));  // k_match packs the train index into 23 bits
  if (npairs == 0) return DVS_OK;
  DVS_HIP(hipSetDevice(m->device));
  const bool few = (long long)npairs * q_stride_rows <= 16384;  // a few jobs: favour wavefront count over per-wave efficiency
  if (few) {
    launch_match_few(m->stream, dim3((q_stride_rows + 63) / 64, npairs), (const u64*)d_q, d_nq, 0, q_stride_rows, (const u64*)d_t, d_nt, 0, t_stride_rows,
                     d_idx, d_dist, t_stride_rows);
    DVS_HIP(hipGetLastError());
    return DVS_OK;
  }
  // many large jobs: the contraction runs on the matrix cores, operands built in the kernel (16-byte row loads: other bases take k_match)
  if (m->use_mfma && t_stride_rows > 0 && (((uintptr_t)d_q | (uintptr_t)d_t) & 15) == 0) {
    launch_match_fp4(m->stream, npairs, d_q, d_nq, q_stride_rows, d_t, d_nt, t_stride_rows, nullptr, nullptr, d_idx, d_dist);
    DVS_HIP(hipGetLastError());
    return DVS_OK;
  }
  constexpr int kSplit = 8, kQPL = 2;
  dim3 grid((q_stride_rows + 64 * kQPL - 1) / (64 * kQPL), npairs);
  hipLaunchKernelGGL((k_match<kSplit, kQPL>), grid, dim3(64 * kSplit), 0, m->stream, (const u64*)d_q, d_nq, 0, q_stride_rows, (const u64*)d_t, d_nt, 0,
                     t_stride_rows, d_idx, d_dist);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status dvs_match_hamming_sequence_device(dvs_matcher* m, const uint8_t* d_desc, const int32_t* d_n, int32_t stride_rows,
                                             int32_t nframes, const uint8_t* d_prev_desc, const int32_t* d_prev_n, int32_t* d_idx,
                                             int32_t* d_dist) {
  DVS_ARG(m && d_desc && d_n && d_idx && d_dist && nframes >= 0 && stride_rows > 0 && stride_rows < (1 << 23));
  DVS_ARG((d_prev_desc == nullptr) == (d_prev_n == nullptr));
  if (nframes == 0) return DVS_OK;
  DVS_HIP(hipSetDevice(m->device));
  if (!d_prev_desc) {  // no predecessor: frame 0 matches against nothing (train_idx -1, dist INT32_MAX)
    if (!m->d_zero) {
      DVS_HIP(hipMalloc((void**)&m->d_zero, 64));
      // on the matcher's own (non-blocking) stream, i.e. ordered before the kernel below: a hipMemset on the null stream is not,
      // and the kernel then read an uninitialised row count (intermittent GPU fault in tests/test_gpu_match.py)
      DVS_HIP(hipMemsetAsync(m->d_zero, 0, 64, m->stream));
    }
    d_prev_desc = (const uint8_t*)m->d_zero; d_prev_n = (const int32_t*)m->d_zero;
  }
  // from 7 frames of 2 000 descriptors on (the four-stream schedule's match stream; up to 6 frames a lane matches from LDS, k_match_lds):
  // with the operands built in the kernel the matrix-core match needs no second launch and wins earlier than rounds 2-4's 16 384 rows
  // (8 frames per step 82.2 -> 90 k frames/s, 7: 77.2 -> 81.5 k; 6 frames on a lane: no gain)
  constexpr long long kMfmaMinRows = 14000;
  if (m->use_mfma && (long long)nframes * stride_rows > kMfmaMinRows && (((uintptr_t)d_desc | (uintptr_t)d_prev_desc) & 15) == 0) {
    // job p = frame p against frame p - 1 (job 0: the predecessor block): train base shifted back by one frame, never dereferenced for job 0
    launch_match_fp4(m->stream, nframes, d_desc, d_n, stride_rows, d_desc - (size_t)stride_rows * 32, d_n - 1, stride_rows, d_prev_desc, d_prev_n, d_idx, d_dist);
    DVS_HIP(hipGetLastError());
    return DVS_OK;
  }
  // a few jobs (everything below the matrix-core threshold): favour wavefront count over per-wave efficiency, as the batch entry point does —
  // one 2000 x 2000 job on <8, 2> occupies 16 workgroups for 39 us
  // train of job p >= 1 = frame p - 1: the base pointers are shifted back by one frame and never dereferenced for job 0
  if ((long long)nframes * stride_rows <= 16384)
    launch_match_few(m->stream, dim3((stride_rows + 63) / 64, nframes), (const u64*)d_desc, d_n, 0, stride_rows,
                     (const u64*)(d_desc - (size_t)stride_rows * 32), d_n - 1, 0, stride_rows, d_idx, d_dist, stride_rows, (const u64*)d_prev_desc, d_prev_n);
  else   // DVS_MATCH_MFMA=0 with many jobs: the throughput shape
    hipLaunchKernelGGL((k_match<8, 2>), dim3((stride_rows + 127) / 128, nframes), dim3(512), 0, m->stream, (const u64*)d_desc, d_n, 0, stride_rows,
                       (const u64*)(d_desc - (size_t)stride_rows * 32), d_n - 1, 0, stride_rows, d_idx, d_dist, (const u64*)d_prev_desc, d_prev_n);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status dvs_match_hamming(dvs_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t* train_idx,
                             int32_t* dist) {
  DVS_ARG(m && nq >= 0 && nt >= 0);
  if (nq == 0) return DVS_OK;  // empty query -> empty result
  DVS_ARG(q && train_idx && dist && (t || nt == 0));
  DVS_ARG(nt < (1 << 23));  // k_match packs the train index into 23 bits
  DVS_HIP(hipSetDevice(m->device));
  DVS_TRY(grow(&m->d_q, &m->cq, (size_t)nq * 32));
  DVS_TRY(grow(&m->d_t, &m->ct, (size_t)std::max(nt, 1) * 32));
  DVS_TRY(grow(&m->d_idx, &m->cidx, (size_t)nq * 8));
  int* d_idx = (int*)m->d_idx;
  int* d_dist = d_idx + nq;
  // frame-sized sets go through the pinned block (import / export kernels, polled sequence number): no copy commands, no stream wait
  const bool pinned_io = nq <= 16384 && nt <= 16384;
  uint8_t* hio = nullptr; int *hseq = nullptr, *counter = nullptr;
  if (pinned_io) {
    DVS_TRY(matcher_pinned(m, (size_t)(nq + nt) * 32 + (size_t)nq * 8, (void**)&hio, &hseq, &counter));
    memcpy(hio, q, (size_t)nq * 32);
    if (nt) memcpy(hio + (size_t)nq * 32, t, (size_t)nt * 32);
    hipLaunchKernelGGL(k_io_import, dim3((nq * 8 + 255) / 256), dim3(256), 0, m->stream, (const uint32_t*)hio, (uint32_t*)m->d_q, nq * 8);
    if (nt) hipLaunchKernelGGL(k_io_import, dim3((nt * 8 + 255) / 256), dim3(256), 0, m->stream, (const uint32_t*)(hio + (size_t)nq * 32), (uint32_t*)m->d_t, nt * 8);
  } else {
    DVS_HIP(hipMemcpyAsync(m->d_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    if (nt) DVS_HIP(hipMemcpyAsync(m->d_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, m->stream));
  }
  if (nq <= 16384)
    launch_match_few(m->stream, dim3((nq + 63) / 64, 1), (const u64*)m->d_q, (const int*)nullptr, nq, nq, (const u64*)m->d_t, (const int*)nullptr, nt, nt,
                     d_idx, d_dist, nt);
  else
    hipLaunchKernelGGL((k_match<8, 2>), dim3((nq + 127) / 128, 1), dim3(512), 0, m->stream, (const u64*)m->d_q, (const int*)nullptr, nq, nq,
                     (const u64*)m->d_t, (const int*)nullptr, nt, nt, d_idx, d_dist);
  DVS_HIP(hipGetLastError());
  if (pinned_io) {
    uint8_t* hout = hio + (size_t)(nq + nt) * 32;
    const int seq = ++*counter;
    hipLaunchKernelGGL(k_io_export, dim3(1), dim3(256), 0, m->stream, (const uint32_t*)d_idx, (uint32_t*)hout, nq * 2, hseq, seq);
    DVS_HIP(hipGetLastError());
    DVS_TRY(io_wait(hseq, seq, m->stream));
    memcpy(train_idx, hout, (size_t)nq * 4); memcpy(dist, hout + (size_t)nq * 4, (size_t)nq * 4);
    return DVS_OK;
  }
  DVS_HIP(hipMemcpyAsync(train_idx, d_idx, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
  DVS_HIP(hipMemcpyAsync(dist, d_dist, (size_t)nq * 4, hipMemcpyDeviceToHost, m->stream));
  DVS_HIP(hipStreamSynchronize(m->stream));
  return DVS_OK;
}

dvs_status dvs_match_hamming_thresh(dvs_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, int32_t max_dist,
                                    int32_t* pairs, int32_t cap, int32_t* n_pairs) {
  DVS_ARG(m && n_pairs && nq >= 0 && nt >= 0 && cap >= 0);
  *n_pairs = 0;
  if (nq == 0 || nt == 0) return DVS_OK;
  DVS_ARG(q && t && (pairs || cap == 0));
  const long long* d_offs; const int* d_pairs; long long total = 0;
  DVS_TRY(matcher_thresh_device(m, q, nq, t, nt, max_dist, &d_offs, &d_pairs, &total));
  const long long nw = std::min<long long>(total, cap);
  if (nw) DVS_HIP(hipMemcpy(pairs, d_pairs, (size_t)nw * 12, hipMemcpyDeviceToHost));
  *n_pairs = (int32_t)std::min<long long>(total, INT_MAX);
  return DVS_OK;
}

}  // extern "C"
