// ba.hip — sliding-window bundle-adjustment evaluation + solve (boundary B3, include/dvslam_hip.h).
// Replaces, for reference include/dynamic_visual_slam/bundle_adjustment.hpp:
//   WeightedSquaredReprojectionError::operator() + ceres::AutoDiffCostFunction<.,2,4,3,3>   :469-593
//   ceres::HuberLoss(1.345) + corrector, ceres::EigenQuaternionManifold tangent projection      :777, :818
//   ceres::Solve(LEVENBERG_MARQUARDT, SPARSE_SCHUR, ...) as configured at                      :839-851
//
// Device work (FP64, nothing here is a dense contraction -> no MFMA):
//   k_ba_eval    one thread per observation, observations grouped by camera in chunks of <= 256:
//                analytic d r / d(q,t,X) including the quaternion normalisation inside
//                ceres::QuaternionRotatePoint, x the 4x3 plus-Jacobian of EigenQuaternionManifold on the
//                raw (w,x,y,z) memory, Huber corrector; per-chunk fixed-order reduction of the camera's
//                H_pp (21 unique) / g_p (6) / cost through wave shuffles + LDS  -> partials
//   k_ba_reduce  thread per landmark walks its observations in fixed order (H_ll, g_l); further blocks
//                fold the chunk partials per camera and the total cost in chunk order.
// Every reduction has a fixed order, so cost / gradient are bit-reproducible run to run.
// dvs_ba_solve: host LM around the two launches (reduced camera system <= 6K x 6K); dvs_ba_solve_device: the k_lm_* kernels.
#include <float.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <chrono>
#include <vector>
#include "common.h"

namespace dvs {

struct BaChunk { int cam, start, count, pad; };

struct BaDev {
  const double *q, *t, *X, *uv;
  const int *cam, *lm;
  const unsigned char *pose_fixed, *lm_fixed;
  double fx, fy, cx, cy, inv_sigma, huber_a;
  const int* gate;   // if set and *gate == 0 the evaluation launches do nothing (speculative launches of dvs_ba_solve_device)
  // dvs_ba_solve_device, after an accepted step: acc_blocks extra workgroups of k_ba_eval store the evaluated point (q, t, X) as the
  // point of the next iteration (was a launch of its own in front of the evaluation; neither writes what the other reads)
  double *acc_q0, *acc_t0, *acc_X0;
  int acc_blocks, acc_K, acc_L;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o);
  return v;
}

// flags: 1 = store residuals/Jacobians (local, loss-corrected), 2 = store W, 4 = store raw functor outputs
__device__ __forceinline__ void ba_eval_body(const int bid, BaDev P, const BaChunk* __restrict__ chunks, int flags,
                                                 double* __restrict__ res, double* __restrict__ Jp, double* __restrict__ Jl,
                                                 double* __restrict__ W, double* __restrict__ partial,
                                                 double* __restrict__ rawRes, double* __restrict__ rawJq,
                                                 double* __restrict__ rawJt, double* __restrict__ rawJX) {
  __shared__ double wred[4][28];
  __shared__ double stage[256 * 12];  // 24 KB: store staging (see below)
  const BaChunk ch = chunks[bid];
  const int tid = threadIdx.x;
  const int c = ch.cam;
  const bool act = tid < ch.count;
  const int p = ch.start + (act ? tid : 0);
  // camera block is uniform over the workgroup
  const double q0 = P.q[4 * c], q1 = P.q[4 * c + 1], q2 = P.q[4 * c + 2], q3 = P.q[4 * c + 3];
  const double t0 = P.t[3 * c], t1 = P.t[3 * c + 1], t2 = P.t[3 * c + 2];
  const int l = P.lm[p];
  const double X0 = P.X[3 * l], X1 = P.X[3 * l + 1], X2 = P.X[3 * l + 2];
  // ceres::QuaternionRotatePoint: normalise, then p = X + u0*uv + u_v x uv with uv = 2 (u_v x X)
  const double s = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const double u0 = s * q0, u1 = s * q1, u2 = s * q2, u3 = s * q3;
  double uv0 = u2 * X2 - u3 * X1, uv1 = u3 * X0 - u1 * X2, uv2 = u1 * X1 - u2 * X0;
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  double pc0 = X0 + u0 * uv0, pc1 = X1 + u0 * uv1, pc2 = X2 + u0 * uv2;
  pc0 += u2 * uv2 - u3 * uv1; pc1 += u3 * uv0 - u1 * uv2; pc2 += u1 * uv1 - u2 * uv0;
  pc0 += t0; pc1 += t1; pc2 += t2;

  double r[2] = {0, 0};
  double jq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, jt[6] = {0, 0, 0, 0, 0, 0}, jx[6] = {0, 0, 0, 0, 0, 0};
  if (act && !(pc2 <= 0.1)) {  // bundle_adjustment.hpp:545-550: residual 0 and (autodiff) zero Jacobians otherwise
    const double iz = 1.0 / pc2;
    // ceres::Jet division is f.a * (1 / g.a); a cost-only evaluation (no Jacobians requested) runs the functor on plain
    // doubles and divides.  The two differ in the last bit, and Ceres' step test compares exactly these two values.
    const bool jets = flags != 0;
    const double px = (jets ? (P.fx * pc0) * iz : P.fx * pc0 / pc2) + P.cx;
    const double py = (jets ? (P.fy * pc1) * iz : P.fy * pc1 / pc2) + P.cy;
    r[0] = P.inv_sigma * (px - P.uv[2 * p]);
    r[1] = P.inv_sigma * (py - P.uv[2 * p + 1]);
    if (jets) {   // a cost-only evaluation (a trust-region candidate) needs none of the derivatives
    // d r / d p_c
    const double a00 = P.inv_sigma * P.fx * iz, a02 = -P.inv_sigma * P.fx * pc0 * iz * iz;
    const double a11 = P.inv_sigma * P.fy * iz, a12 = -P.inv_sigma * P.fy * pc1 * iz * iz;
    jt[0] = a00; jt[1] = 0; jt[2] = a02; jt[3] = 0; jt[4] = a11; jt[5] = a12;
    // d p_c / d X = R(u)
    const double R00 = 1 - 2 * (u2 * u2 + u3 * u3), R01 = 2 * (u1 * u2 - u0 * u3), R02 = 2 * (u1 * u3 + u0 * u2);
    const double R10 = 2 * (u1 * u2 + u0 * u3), R11 = 1 - 2 * (u1 * u1 + u3 * u3), R12 = 2 * (u2 * u3 - u0 * u1);
    const double R20 = 2 * (u1 * u3 - u0 * u2), R21 = 2 * (u2 * u3 + u0 * u1), R22 = 1 - 2 * (u1 * u1 + u2 * u2);
    jx[0] = a00 * R00 + a02 * R20; jx[1] = a00 * R01 + a02 * R21; jx[2] = a00 * R02 + a02 * R22;
    jx[3] = a11 * R10 + a12 * R20; jx[4] = a11 * R11 + a12 * R21; jx[5] = a11 * R12 + a12 * R22;
    // d p_c / d u  (3 x 4): column 0 = uv ; columns 1..3 = -2 u0 [X]x + 2 (u_v . X) I + 2 u_v X^T - 4 X u_v^T
    const double dX = u1 * X0 + u2 * X1 + u3 * X2;
    double D[3][4];
    D[0][0] = uv0; D[1][0] = uv1; D[2][0] = uv2;
    const double uvv[3] = {u1, u2, u3}, Xv[3] = {X0, X1, X2};
    // [X]x = [[0,-X2,X1],[X2,0,-X0],[-X1,X0,0]]
    const double Xx[3][3] = {{0, -X2, X1}, {X2, 0, -X0}, {-X1, X0, 0}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        D[i][1 + j] = -2 * u0 * Xx[i][j] + (i == j ? 2 * dX : 0.0) + 2 * uvv[i] * Xv[j] - 4 * Xv[i] * uvv[j];
    // chain through u = q / |q| : d u / d q = s (I - u u^T)
    const double uu[4] = {u0, u1, u2, u3};
    double Dq[3][4];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double du = D[i][0] * u0 + D[i][1] * u1 + D[i][2] * u2 + D[i][3] * u3;
#pragma unroll
      for (int j = 0; j < 4; j++) Dq[i][j] = s * (D[i][j] - du * uu[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      jq[j] = a00 * Dq[0][j] + a02 * Dq[2][j];
      jq[4 + j] = a11 * Dq[1][j] + a12 * Dq[2][j];
    }
    }
  }
  if (act && (flags & 4)) {
    if (rawRes) { rawRes[2 * p] = r[0]; rawRes[2 * p + 1] = r[1]; }
    if (rawJq) for (int i = 0; i < 8; i++) rawJq[8 * p + i] = jq[i];
    if (rawJt) for (int i = 0; i < 6; i++) rawJt[6 * p + i] = jt[i];
    if (rawJX) for (int i = 0; i < 6; i++) rawJX[6 * p + i] = jx[i];
  }
  // tangent projection: J_rot = J_q (2x4) * PlusJacobian(4x3) of EigenQuaternionManifold evaluated on raw memory
  // x = (q0,q1,q2,q3) read as (x,y,z,w):  [ x3, x2,-x1; -x2, x3, x0; x1,-x0, x3; -x0,-x1,-x2 ]
  double jp[12];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const double* a = &jq[4 * k];
    jp[6 * k + 0] = a[0] * q3 - a[1] * q2 + a[2] * q1 - a[3] * q0;
    jp[6 * k + 1] = a[0] * q2 + a[1] * q3 - a[2] * q0 - a[3] * q1;
    jp[6 * k + 2] = -a[0] * q1 + a[1] * q0 + a[2] * q3 - a[3] * q2;
    jp[6 * k + 3] = jt[3 * k]; jp[6 * k + 4] = jt[3 * k + 1]; jp[6 * k + 5] = jt[3 * k + 2];
  }
  // Huber loss + corrector (rho'' <= 0  =>  r, J scaled by sqrt(rho'))
  const double sq = r[0] * r[0] + r[1] * r[1];
  const double b = P.huber_a * P.huber_a;
  double rho0 = sq, rho1 = 1.0;
  if (sq > b) {
    const double rr = sqrt(sq);
    rho0 = 2.0 * P.huber_a * rr - b;
    rho1 = fmax(DBL_MIN, P.huber_a / rr);
  }
  if (flags == 0) {
    // cost only: one value through the reduction below instead of 28 — the same association as the butterfly's (pairwise by lane
    // bits 5 .. 0, then the four wavefronts in order), so the same bits
    double cst = act ? 0.5 * rho0 : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cst += __shfl_xor(cst, o);
    if ((tid & 63) == 0) wred[tid >> 6][27] = cst;
    __syncthreads();
    if (tid == 0) partial[(size_t)bid * 28 + 27] = ((wred[0][27] + wred[1][27]) + wred[2][27]) + wred[3][27];
    return;
  }
  const double sc = sqrt(rho1);
  r[0] *= sc; r[1] *= sc;
#pragma unroll
  for (int i = 0; i < 12; i++) jp[i] *= sc;
#pragma unroll
  for (int i = 0; i < 6; i++) jx[i] *= sc;
  // Stores go through LDS: a thread's 2 + 12 + 6 + 18 doubles are array-of-structures records (stride 16 / 96 / 48 / 144 bytes
  // across lanes — every store instruction would touch 64 cache lines), but the workgroup's <= 256 consecutive observations
  // form ONE contiguous block of each array, which is then written out with consecutive 8-byte lanes.
  const bool pf = P.pose_fixed[c] != 0;
  {
    double* stg = stage;
    auto flush = [&](double* dst, int n) {  // n doubles per observation staged at stg[tid * n ..]; barriers on both sides
      __syncthreads();
      const int tot = ch.count * n;
      double* o = dst + (size_t)ch.start * n;
      for (int k = tid; k < tot; k += 256) o[k] = stg[k];
      __syncthreads();
    };
    if (flags & 1) {
      if (act) {
#pragma unroll
        for (int i = 0; i < 12; i++) stg[12 * tid + i] = jp[i];
      }
      flush(Jp, 12);
      if (act) {
#pragma unroll
        for (int i = 0; i < 6; i++) stg[8 * tid + i] = jx[i];
        stg[8 * tid + 6] = r[0]; stg[8 * tid + 7] = r[1];
      }
      __syncthreads();
      {  // Jl (6 per observation) and res (2 per observation) from the 8-double records
        const int tot6 = ch.count * 6, tot2 = ch.count * 2;
        double* o6 = Jl + (size_t)ch.start * 6;
        double* o2 = res + (size_t)ch.start * 2;
        for (int k = tid; k < tot6; k += 256) { const int e = k / 6; o6[k] = stg[8 * e + (k - 6 * e)]; }
        for (int k = tid; k < tot2; k += 256) o2[k] = stg[8 * (k >> 1) + 6 + (k & 1)];
      }
      __syncthreads();
    }
    if (flags & 2) {
      const bool zero = pf || P.lm_fixed[l] != 0;
      for (int half = 0; half < 2; half++) {  // W is 18 doubles per observation: 128 observations per pass fit the buffer
        if (act && (tid >> 7) == half) {
          double* sw = stg + 18 * (tid & 127);
#pragma unroll
          for (int a = 0; a < 6; a++)
#pragma unroll
            for (int bb = 0; bb < 3; bb++) sw[3 * a + bb] = zero ? 0.0 : jp[a] * jx[bb] + jp[6 + a] * jx[3 + bb];
        }
        __syncthreads();
        const int first = 128 * half, cntp = min(max(ch.count - first, 0), 128);
        double* o = W + (size_t)(ch.start + first) * 18;
        for (int k = tid; k < cntp * 18; k += 256) o[k] = stg[k];
        __syncthreads();
      }
    }
  }
  // camera partials: 21 unique H_pp entries (row-major upper triangle), 6 gradient entries, cost
  double v[28];
  {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int bb = a; bb < 6; bb++) v[k++] = (act && !pf) ? jp[a] * jp[bb] + jp[6 + a] * jp[6 + bb] : 0.0;
#pragma unroll
    for (int a = 0; a < 6; a++) v[21 + a] = (act && !pf) ? jp[a] * r[0] + jp[6 + a] * r[1] : 0.0;
    v[27] = act ? 0.5 * rho0 : 0.0;
  }
  const int w = tid >> 6, lane = tid & 63;
  // 28 wave-wide sums by a TRANSPOSING butterfly: at the step with lane bit B a lane keeps the half of the values its bit B
  // selects and adds its partner's copies of them, so the cross-lane traffic halves each step — 32 exchanges instead of the
  // 28 x 6 of one shuffle tree per value (which was 54 % of this kernel's time).  Fixed association: pairwise by lane bits
  // 5, 4, 3, 2, 1, 0.  Value k ends up in the lanes whose bits 5..1 spell k.
  {
    double q[32];
#pragma unroll
    for (int k = 0; k < 32; k++) q[k] = k < 28 ? v[k] : 0.0;
#define DVS_BFLY64(n, o)                                      \
    {                                                         \
      const bool hi = (lane & (o)) != 0;                      \
      _Pragma("unroll") for (int j = 0; j < (n); j++) {       \
        const double keep = hi ? q[(n) + j] : q[j];           \
        const double send = hi ? q[j] : q[(n) + j];           \
        q[j] = keep + __shfl_xor(send, (o));                  \
      }                                                       \
    }
    DVS_BFLY64(16, 32) DVS_BFLY64(8, 16) DVS_BFLY64(4, 8) DVS_BFLY64(2, 4) DVS_BFLY64(1, 2)
#undef DVS_BFLY64
    const double tot = q[0] + __shfl_xor(q[0], 1);
    const int k = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0 && k < 28) wred[w][k] = tot;
  }
  __syncthreads();
  if (tid < 28) partial[(size_t)bid * 28 + tid] = ((wred[0][tid] + wred[1][tid]) + wred[2][tid]) + wred[3][tid];
}

__global__ __launch_bounds__(256) void k_ba_eval(BaDev P, const BaChunk* __restrict__ chunks, int flags,
                                                 double* __restrict__ res, double* __restrict__ Jp, double* __restrict__ Jl,
                                                 double* __restrict__ W, double* __restrict__ partial,
                                                 double* __restrict__ rawRes, double* __restrict__ rawJq,
                                                 double* __restrict__ rawJt, double* __restrict__ rawJX) {
  if (P.gate && !*P.gate) return;
  const int nEval = (int)gridDim.x - P.acc_blocks;
  if ((int)blockIdx.x >= nEval) {
    const int i0 = ((int)blockIdx.x - nEval) * 256 + (int)threadIdx.x, stride = P.acc_blocks * 256;
    for (int i = i0; i < 4 * P.acc_K; i += stride) P.acc_q0[i] = P.q[i];
    for (int i = i0; i < 3 * P.acc_K; i += stride) P.acc_t0[i] = P.t[i];
    for (int i = i0; i < 3 * P.acc_L; i += stride) P.acc_X0[i] = P.X[i];
    return;
  }
  ba_eval_body((int)blockIdx.x, P, chunks, flags, res, Jp, Jl, W, partial, rawRes, rawJq, rawJt, rawJX);
}

__device__ __forceinline__ void ba_reduce_body(const int bid, const int nReduce, BaDev P, int K, int L, int nChunks, const BaChunk* __restrict__ chunks,
                                                   const int* __restrict__ camChunkStart, const int* __restrict__ lmStart,
                                                   const int* __restrict__ lmObs, const double* __restrict__ res,
                                                   const double* __restrict__ Jl, const double* __restrict__ partial, int lmBlocks,
                                                   int withLm, int costOnly, double* __restrict__ Hpp, double* __restrict__ Hll,
                                                   double* __restrict__ g, double* __restrict__ cost, double* __restrict__ costCam,
                                                   int* __restrict__ ticketCounter) {
  const int tid = threadIdx.x;
  if ((int)bid < lmBlocks) {
    if (!withLm) return;
    const int l = bid * 256 + tid;
    if (l >= L) return;
    double h[6] = {0, 0, 0, 0, 0, 0}, gl[3] = {0, 0, 0};
    if (!P.lm_fixed[l]) {
      auto acc = [&](const double* j, double r0, double r1) {
        h[0] += j[0] * j[0] + j[3] * j[3]; h[1] += j[0] * j[1] + j[3] * j[4]; h[2] += j[0] * j[2] + j[3] * j[5];
        h[3] += j[1] * j[1] + j[4] * j[4]; h[4] += j[1] * j[2] + j[4] * j[5]; h[5] += j[2] * j[2] + j[5] * j[5];
        gl[0] += j[0] * r0 + j[3] * r1; gl[1] += j[1] * r0 + j[4] * r1; gl[2] += j[2] * r0 + j[5] * r1;
      };
      // a landmark's observations sit in K different cameras' blocks: every record is its own L2 round trip.  Four records
      // (index, then 8 doubles each) are requested together instead of one after the other; the sums keep their order.
      int e = lmStart[l];
      const int e1 = lmStart[l + 1];
      for (; e + 4 <= e1; e += 4) {
        int pp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) pp[u] = lmObs[e + u];
        double jj[4][6], rr[4][2];
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
          for (int i = 0; i < 6; i++) jj[u][i] = Jl[6 * (size_t)pp[u] + i];
          rr[u][0] = res[2 * (size_t)pp[u]]; rr[u][1] = res[2 * (size_t)pp[u] + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc(jj[u], rr[u][0], rr[u][1]);
      }
      for (; e < e1; e++) {
        const int p = lmObs[e];
        double j6[6];
#pragma unroll
        for (int i = 0; i < 6; i++) j6[i] = Jl[6 * (size_t)p + i];
        acc(j6, res[2 * (size_t)p], res[2 * (size_t)p + 1]);
      }
    }
    double* H = Hll + 9 * (size_t)l;
    H[0] = h[0]; H[1] = h[1]; H[2] = h[2]; H[3] = h[1]; H[4] = h[3]; H[5] = h[4]; H[6] = h[2]; H[7] = h[4]; H[8] = h[5];
    g[6 * K + 3 * l] = gl[0]; g[6 * K + 3 * l + 1] = gl[1]; g[6 * K + 3 * l + 2] = gl[2];
    return;
  }
  // camera fold: 8 cameras per workgroup, thread (c, k) sums the camera's chunk partials in chunk order (k = 27: cost)
  const int cb = (int)bid - lmBlocks;
  const int c = cb * 8 + (tid >> 5), k = tid & 31;
  // costOnly (a trust-region candidate's cost, flags == 0): fold only the cost column — H_pp and g keep the ACCEPTED point's
  // values, which the next trial step needs again if this candidate is rejected
  if (c < K && k < 28 && (!costOnly || k == 27)) {
    double sacc = 0;
    for (int ch = camChunkStart[c]; ch < camChunkStart[c + 1]; ch++) sacc += partial[(size_t)ch * 28 + k];
    if (k < 21) {
      int a = 0, rem = k;
      while (rem >= 6 - a) { rem -= 6 - a; a++; }
      const int bcol = a + rem;
      Hpp[36 * (size_t)c + 6 * a + bcol] = sacc;
      Hpp[36 * (size_t)c + 6 * bcol + a] = sacc;
    } else if (k < 27) {
      g[6 * c + (k - 21)] = sacc;
    } else {
      costCam[c] = sacc;
    }
  }
  // total cost = sum over cameras in index order, done by whichever camera workgroup arrives last (agent-scope
  // release -> ticket -> acquire, cdna_hip_programming.md Guideline 16); the ticket counter is reset for the next launch
  __shared__ int s_last;
  __shared__ double s_red[256];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int nCamBlocks = nReduce - lmBlocks;
    const int ticket = __hip_atomic_fetch_add(ticketCounter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == nCamBlocks - 1 ? 1 : 0;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (s_last) {
    double sacc = 0;
    const int per = (K + 255) / 256;
    for (int i = tid * per; i < min(K, (tid + 1) * per); i++) sacc += __hip_atomic_load(&costCam[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_red[tid] = sacc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {  // fixed-shape tree: deterministic
      if (tid < o) s_red[tid] += s_red[tid + o];
      __syncthreads();
    }
    if (tid == 0) { *cost = s_red[0]; __hip_atomic_store(ticketCounter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
}

__global__ __launch_bounds__(256) void k_ba_reduce(BaDev P, int K, int L, int nChunks, const BaChunk* __restrict__ chunks,
                                                   const int* __restrict__ camChunkStart, const int* __restrict__ lmStart,
                                                   const int* __restrict__ lmObs, const double* __restrict__ res,
                                                   const double* __restrict__ Jl, const double* __restrict__ partial, int lmBlocks,
                                                   int withLm, int costOnly, double* __restrict__ Hpp, double* __restrict__ Hll,
                                                   double* __restrict__ g, double* __restrict__ cost, double* __restrict__ costCam,
                                                   int* __restrict__ ticketCounter) {
  if (P.gate && !*P.gate) return;
  ba_reduce_body((int)blockIdx.x, (int)gridDim.x, P, K, L, nChunks, chunks, camChunkStart, lmStart, lmObs, res, Jl, partial, lmBlocks, withLm, costOnly, Hpp, Hll, g, cost, costCam, ticketCounter);
}


// =============================================================================================================================
// Device-resident Levenberg-Marquardt step: the linear algebra of dvs_ba_solve — Jacobi scaling, LM
// diagonal, landmark elimination (Schur complement), the reduced camera system's Cholesky, back-substitution, the model cost
// change and the candidate point — as kernels over the buffers k_ba_eval / k_ba_reduce leave in HBM.  The host keeps only the
// trust-region decisions and reads one 64-byte status record per trial step.  All reductions run in a fixed order.
// =============================================================================================================================
struct LmStatus { int ok, finite; double model_change, sn, xn, cand_cost, gmax, x_cost; int seq, accept; };   // seq: number of this publication; accept: the trial step's verdict (k_lm_norms)

__device__ __forceinline__ double block_sum_fixed(double v, double* sm) {  // 256 threads, fixed tree
  const int tid = threadIdx.x;
  __syncthreads();
  sm[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) sm[tid] += sm[tid + s];
    __syncthreads();
  }
  return sm[0];
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

__device__ __forceinline__ void quat_plus_dev(const double* x, const double* d, double* o) {  // EigenQuaternionManifold::Plus
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { for (int i = 0; i < 4; i++) o[i] = x[i]; return; }
  const double sn = sin(nd) / nd;
  const double dx = sn * d[0], dy = sn * d[1], dz = sn * d[2], dw = cos(nd);
  o[3] = dw * x[3] - dx * x[0] - dy * x[1] - dz * x[2];
  o[0] = dw * x[0] + dx * x[3] + dy * x[2] - dz * x[1];
  o[1] = dw * x[1] + dy * x[3] + dz * x[0] - dx * x[2];
  o[2] = dw * x[2] + dz * x[3] + dx * x[1] - dy * x[0];
}

// Jacobi scaling 1 / (1 + sqrt(H_jj)), fixed at the first Jacobian
__global__ void k_lm_scale(int K, int L, const double* __restrict__ Hpp, const double* __restrict__ Hll, double* __restrict__ scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < 6 * K) { const int c = j / 6, a = j - 6 * c; scale[j] = 1.0 / (1.0 + sqrt(Hpp[36 * (size_t)c + 7 * a])); }
  else if (j < 6 * K + 3 * L) { const int k = j - 6 * K, l = k / 3, a = k - 3 * l; scale[j] = 1.0 / (1.0 + sqrt(Hll[9 * (size_t)l + 4 * a])); }
}

// per observation: the Jacobi-scaled W block and Y = W V^-1 of its landmark — every observation inverts its landmark's 3 x 3 block
// itself (the same arithmetic as k_lm_landmarks, ~60 flops) instead of waiting for a launch that does it once per landmark; the
// landmark's first observation stores Vinv and the refreshed LM diagonal.  Threads R .. R + L + K - 1 refresh the diagonal of the
// cameras and of the landmarks nobody observes.
__global__ __launch_bounds__(256) void k_lm_observations(int K, int L, int R, const double* __restrict__ Hpp, const double* __restrict__ Hll,
                                                         const double* __restrict__ W, const int* __restrict__ cam, const int* __restrict__ lm,
                                                         const int* __restrict__ lmStart, const int* __restrict__ lmObs,
                                                         const double* __restrict__ scale, double* __restrict__ diag,
                                                         const unsigned char* __restrict__ active, double radius, int refresh,
                                                         double* __restrict__ Vinv, double* __restrict__ Ws, double* __restrict__ Y,
                                                         LmStatus* __restrict__ st) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= R) {
    const int i = p - R;
    if (!refresh) return;
    if (i < L) {
      if (lmStart[i + 1] == lmStart[i]) {
        const int j0 = 6 * K + 3 * i;
        for (int a = 0; a < 3; a++) diag[j0 + a] = clampd(Hll[9 * (size_t)i + 4 * a] * scale[j0 + a] * scale[j0 + a], 1e-6, 1e32);
      }
    } else if (i - L < K) {
      const int c = i - L;
      for (int a = 0; a < 6; a++) { const int j = 6 * c + a; diag[j] = clampd(Hpp[36 * (size_t)c + 7 * a] * scale[j] * scale[j], 1e-6, 1e32); }
    }
    return;
  }
  const int l = lm[p], c = cam[p], j0 = 6 * K + 3 * l;
  const bool first = lmObs[lmStart[l]] == p;
  double dg[3];
  for (int a = 0; a < 3; a++) dg[a] = refresh ? clampd(Hll[9 * (size_t)l + 4 * a] * scale[j0 + a] * scale[j0 + a], 1e-6, 1e32) : diag[j0 + a];
  if (first && refresh) for (int a = 0; a < 3; a++) diag[j0 + a] = dg[a];
  if (!active[j0]) return;
  double V[9];
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) V[3 * a + b] = Hll[9 * (size_t)l + 3 * a + b] * scale[j0 + a] * scale[j0 + b];
  for (int a = 0; a < 3; a++) V[4 * a] += dg[a] / radius;
  const double a_ = V[0], b_ = V[1], c_ = V[2], d_ = V[3], e_ = V[4], f_ = V[5], g_ = V[6], h_ = V[7], i_ = V[8];
  const double det = a_ * (e_ * i_ - f_ * h_) - b_ * (d_ * i_ - f_ * g_) + c_ * (d_ * h_ - e_ * g_);
  if (det == 0 || !isfinite(det)) { st->ok = 0; return; }
  const double id = 1.0 / det;
  double Vi[9];
  Vi[0] = (e_ * i_ - f_ * h_) * id; Vi[1] = (c_ * h_ - b_ * i_) * id; Vi[2] = (b_ * f_ - c_ * e_) * id;
  Vi[3] = (f_ * g_ - d_ * i_) * id; Vi[4] = (a_ * i_ - c_ * g_) * id; Vi[5] = (c_ * d_ - a_ * f_) * id;
  Vi[6] = (d_ * h_ - e_ * g_) * id; Vi[7] = (b_ * g_ - a_ * h_) * id; Vi[8] = (a_ * e_ - b_ * d_) * id;
  if (first) for (int k = 0; k < 9; k++) Vinv[9 * (size_t)l + k] = Vi[k];
  for (int a = 0; a < 6; a++) {
    double w[3];
    for (int b = 0; b < 3; b++) { w[b] = W[18 * (size_t)p + 3 * a + b] * scale[6 * c + a] * scale[j0 + b]; Ws[18 * (size_t)p + 3 * a + b] = w[b]; }
    for (int b = 0; b < 3; b++) Y[18 * (size_t)p + 3 * a + b] = w[0] * Vi[b] + w[1] * Vi[3 + b] + w[2] * Vi[6 + b];
  }
}

// 32 wave-wide sums by a transposing butterfly (see k_ba_eval): returns this lane's total; value k lives in the lanes whose bits
// 5..1 spell k (both lanes of the pair hold it).  Fixed association: pairwise by lane bits 5, 4, 3, 2, 1, 0.
__device__ __forceinline__ double wave_butterfly32(double (&q)[32], int lane) {
#define DVS_BFLY64(n, o)                                      \
  {                                                           \
    const bool hi = (lane & (o)) != 0;                        \
    _Pragma("unroll") for (int j = 0; j < (n); j++) {         \
      const double keep = hi ? q[(n) + j] : q[j];             \
      const double send = hi ? q[j] : q[(n) + j];             \
      q[j] = keep + __shfl_xor(send, (o));                    \
    }                                                         \
  }
  DVS_BFLY64(16, 32) DVS_BFLY64(8, 16) DVS_BFLY64(4, 8) DVS_BFLY64(2, 4) DVS_BFLY64(1, 2)
#undef DVS_BFLY64
  return q[0] + __shfl_xor(q[0], 1);
}
__device__ __forceinline__ int butterfly32_index(int lane) {
  return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

// reduced camera system, kSchurSplit workgroups per 6x6 block (ci, ck) of the lower triangle:  S = (H_pp + D/radius) - sum_l Y_l,ci W_l,ck^T,
// rhs likewise.  A thread's landmarks are a chain of dependent loads (observation slots, then 36 doubles); with one workgroup per
// block that was 8 landmarks = 24 us on 81 CUs.  Split s sums the landmarks l = 256 s + thread (mod 256 kSchurSplit) and writes its
// total into S + (s + 1) n^2 / rhs + (s + 1) n; split 0 also writes the H_pp part into S / rhs; k_lm_chol adds them up in split order.
constexpr int kSchurSplit = 4;
__global__ __launch_bounds__(256) void k_lm_schur(int K, int L, int n, const int* __restrict__ slotCam, const int* __restrict__ obsOf,
                                                  const unsigned char* __restrict__ active, const double* __restrict__ Hpp,
                                                  const double* __restrict__ g, const double* __restrict__ scale,
                                                  const double* __restrict__ diag, double radius, const double* __restrict__ Ws,
                                                  const double* __restrict__ Y, double* __restrict__ S, double* __restrict__ rhs) {
  const int ci = blockIdx.x, ck = blockIdx.y, sp = blockIdx.z;
  if (ck > ci) return;                   // the factorisation reads the lower triangle only
  const int cI = slotCam[ci], cK = slotCam[ck];
  double acc[36], r[6];
  for (int k = 0; k < 36; k++) acc[k] = 0.0;
  for (int k = 0; k < 6; k++) r[k] = 0.0;
  for (int l = 256 * sp + threadIdx.x; l < L; l += 256 * kSchurSplit) {
    const int j0 = 6 * K + 3 * l;
    if (!active[j0]) continue;
    const int e = obsOf[(size_t)l * K + cI], f = obsOf[(size_t)l * K + cK];
    if (e < 0 || f < 0) continue;
    const double* y = Y + 18 * (size_t)e;
    const double* w = Ws + 18 * (size_t)f;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) acc[6 * a + b] += y[3 * a] * w[3 * b] + y[3 * a + 1] * w[3 * b + 1] + y[3 * a + 2] * w[3 * b + 2];
    if (ci == ck) {
      const double gl0 = g[j0] * scale[j0], gl1 = g[j0 + 1] * scale[j0 + 1], gl2 = g[j0 + 2] * scale[j0 + 2];
      for (int a = 0; a < 6; a++) r[a] += y[3 * a] * gl0 + y[3 * a + 1] * gl1 + y[3 * a + 2] * gl2;
    }
  }
  // 42 block sums: two wave butterflies (32 + 10 values) per wavefront, then the four wavefronts' totals in wave order
  __shared__ double wtot[4][48];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    double q[32];
#pragma unroll
    for (int k = 0; k < 32; k++) q[k] = acc[k];
    const double t0 = wave_butterfly32(q, lane);
#pragma unroll
    for (int k = 0; k < 32; k++) q[k] = k < 4 ? acc[32 + k] : (k < 10 ? r[k - 4] : 0.0);
    const double t1 = wave_butterfly32(q, lane);
    const int k = butterfly32_index(lane);
    if ((lane & 1) == 0) { wtot[wv][k] = t0; if (k < 10) wtot[wv][32 + k] = t1; }
  }
  __syncthreads();
  double* Sp = S + (size_t)(sp + 1) * n * n;
  double* rp = rhs + (size_t)(sp + 1) * n;
  if (threadIdx.x < 36) {
    const int k = threadIdx.x, a = k / 6, b = k - 6 * a;
    Sp[(size_t)(6 * ci + a) * n + 6 * ck + b] = ((wtot[0][k] + wtot[1][k]) + wtot[2][k]) + wtot[3][k];
    if (sp == 0) {
      double v = 0.0;
      if (ci == ck) {
        v = Hpp[36 * (size_t)cI + 6 * a + b] * scale[6 * cI + a] * scale[6 * cI + b];
        if (a == b) v += diag[6 * cI + a] / radius;
      }
      S[(size_t)(6 * ci + a) * n + 6 * ck + b] = v;
    }
  } else if (ci == ck && threadIdx.x < 42) {
    const int a = threadIdx.x - 36, k = 36 + a;
    rp[6 * ci + a] = ((wtot[0][k] + wtot[1][k]) + wtot[2][k]) + wtot[3][k];
    if (sp == 0) rhs[6 * ci + a] = g[6 * cI + a] * scale[6 * cI + a];
  }
}

// Dense Cholesky + two triangular solves of the n x n reduced system (n = 6 per free camera), one workgroup, everything in LDS.
// Right-looking on the matrix AUGMENTED by the right-hand side as row n, in place, BLOCKED by the 6 columns of a camera (round 4; the
// column-at-a-time form of rounds 1-3 paid one workgroup barrier and three LDS round trips per column: 31.7 us for n = 54):
//   * the first wavefront is the PANEL: lane = row (two rows per lane past 64), the row's six entries of the block column in registers.
//     It factors the 6 x 6 diagonal block and scales the rows below in one go — pivots and the block's l_tj broadcast by readlane, no
//     LDS and no barrier inside the chain of six dependent sqrt / divide steps — and writes the finished columns to LDS;
//   * the other fifteen wavefronts subtract the panel's rank-6 product from the trailing triangle, six mul + sub per element IN COLUMN
//     ORDER, so element (i, k) receives the subtractions j = 0, 1, ... exactly as the host routine's dot products do (chol_solve): the
//     factor is bit-identical to it;
//   * look-ahead: while they do, the panel wavefront applies the same update to the NEXT block column only and factors it straight from
//     its registers.  One barrier per block column (n / 6 + 1 in all).  Measured: 31.7 -> 27.4 us for n = 54 (the chain of 54 dependent
//     f64 sqrt + divide + broadcast steps of ONE wavefront is what is left: ~0.4 us each).  Keeping the next block column in registers as
//     well and subtracting every finished column from it inside the chain — bit-identical — runs 55.8 us: the fully unrolled body with
//     run-time lane indices for the broadcasts and a third register set serialises more than it hides (profiles/r04_ba_lm_kernel_stats.csv).
// Row n undergoes exactly the host's forward substitution (y_j = (b_j - sum_k l_jk y_k) / l_jj, same order), so L y = b costs nothing
// extra.  The backward substitution applies its updates from the last unknown down (a different association from the host's:
// rounding-level) — for n <= 64 in the registers of one wavefront (x_j broadcast by readlane, the factor's row prefetched: no LDS
// round trip and no barrier on the chain of n dependent steps).  Writes the (not yet negated) camera steps.
constexpr int kCholThreads = 1024;   // one panel wavefront + fifteen for the trailing update
__global__ __launch_bounds__(kCholThreads) void k_lm_chol(int K, int n, const int* __restrict__ slotCam, const double* __restrict__ S,
                                                 const double* __restrict__ rhs, double* __restrict__ step, LmStatus* __restrict__ st) {
  extern __shared__ double lds[];
  double* A = lds;                             // lower triangle, row-major (n + 1) x n, factored in place: row n = the right-hand side -> y
  __shared__ int bad;
  const int tid = threadIdx.x;
  // the system from k_lm_schur's pieces: the H_pp part minus the kSchurSplit landmark sums in split order (block lower triangle)
#pragma unroll 4
  for (int r = tid >> 6; r < n; r += kCholThreads / 64) {   // a wavefront per row: no index division, the loads of four rows in flight together
    const int cend = 6 * (r / 6) + 6;
    for (int c = tid & 63; c < n; c += 64) {
      const int i = r * n + c;
      double v = 0.0;
      if (c < cend) {
        double p[kSchurSplit];
#pragma unroll
        for (int sp = 0; sp < kSchurSplit; sp++) p[sp] = S[(size_t)(sp + 1) * n * n + i];
        double tot = p[0];
#pragma unroll
        for (int sp = 1; sp < kSchurSplit; sp++) tot += p[sp];
        v = S[i] - tot;
      }
      A[i] = v;
    }
  }
  for (int i = tid; i < n; i += kCholThreads) {
    double tot = rhs[n + i];
    for (int sp = 1; sp < kSchurSplit; sp++) tot += rhs[(size_t)(sp + 1) * n + i];
    A[(size_t)n * n + i] = rhs[i] - tot;
  }
  for (int i = tid; i < 6 * K; i += kCholThreads) step[i] = 0.0;
  if (tid == 0) bad = 0;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  auto bcast = [](double v, int src) -> double {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  // panel wavefront: a0 / a1 = the six entries of block column c of rows c + lane / c + lane + 64 (all earlier updates applied).
  // Factors the diagonal block, scales the rows below, writes columns c .. c + 5.
  auto panel = [&](int c, double (&a0)[6], double (&a1)[6]) {
    const int r0 = c + lane, r1 = c + lane + 64;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double piv = bcast(a0[j], j);
      if (!(piv > 0) && lane == 0) bad = 1;   // the factorisation runs on (NaNs from here): no flag to poll on the chain of columns
      const double d = sqrt(piv);
      a0[j] = lane == j ? d : a0[j] / d;
      a1[j] = a1[j] / d;
#pragma unroll
      for (int t = j + 1; t < 6; t++) {
        const double lt = bcast(a0[j], t);    // l of row c + t, column c + j
        a0[t] -= a0[j] * lt;
        a1[t] -= a1[j] * lt;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
      if (lane >= j && r0 <= n) A[(size_t)r0 * n + c + j] = a0[j];
      if (r1 <= n) A[(size_t)r1 * n + c + j] = a1[j];
    }
  };
  double a0[6], a1[6];
  if (wave == 0) {
    const int r0 = min(lane, n), r1 = min(lane + 64, n);
#pragma unroll
    for (int j = 0; j < 6; j++) { a0[j] = A[(size_t)r0 * n + j]; a1[j] = A[(size_t)r1 * n + j]; }
    panel(0, a0, a1);
  }
  __syncthreads();
  constexpr int kTy = (kCholThreads - 64) / 16;
  const int t3 = tid - 64, ty = t3 >> 4, tx = t3 & 15;   // the other wavefronts: kTy x 16 tiling of the trailing block
  for (int c0 = 0; c0 < n; c0 += 6) {
    const int c1 = c0 + 6;
    if (wave == 0) {
      if (c1 < n) {
        // look-ahead: block column c1 of rows c1 + lane (+ 64) minus the panel's product, then its factorisation from the registers
        const int r0 = min(c1 + lane, n), r1 = min(c1 + lane + 64, n);
        double l0[6], l1[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
          l0[j] = A[(size_t)r0 * n + c0 + j]; l1[j] = A[(size_t)r1 * n + c0 + j];
          a0[j] = A[(size_t)r0 * n + c1 + j]; a1[j] = A[(size_t)r1 * n + c1 + j];
        }
#pragma unroll
        for (int kk = 0; kk < 6; kk++)
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const double lk = bcast(l0[j], kk);   // l of row c1 + kk, column c0 + j
            a0[kk] -= l0[j] * lk;
            a1[kk] -= l1[j] * lk;
          }
        panel(c1, a0, a1);
      }
    } else {
      const int s0 = c1 + 6;               // first row / column the look-ahead does not cover
      for (int i = s0 + ty; i <= n; i += kTy) {
        double li[6];
#pragma unroll
        for (int j = 0; j < 6; j++) li[j] = A[(size_t)i * n + c0 + j];
        const int kmax = min(i, n - 1);    // the right-hand side row has no diagonal element
        for (int k = s0 + tx; k <= kmax; k += 16) {
          double acc = A[(size_t)i * n + k];
#pragma unroll
          for (int j = 0; j < 6; j++) acc -= li[j] * A[(size_t)k * n + c0 + j];
          A[(size_t)i * n + k] = acc;
        }
      }
    }
    __syncthreads();
  }
  if (bad) { if (tid == 0) st->ok = 0; return; }
  if (tid >= 64) return;
  const double* Lm = A;
  const double* y = A + (size_t)n * n;
  if (n <= 64) {
    // backward: L^T x = y with x in registers (lane i holds unknown i)
    const int li = min(lane, n - 1);
    double bi = y[li];
    // (x_j = b_j * (1 / l_jj): the reciprocal is formed one step ahead, off the chain of n dependent steps — one rounding more than the
    // host's division, in a substitution whose association already differs from the host's)
    double lrow = Lm[(size_t)(n - 1) * n + li], rj = 1.0 / Lm[(size_t)(n - 1) * n + (n - 1)];
    for (int j = n - 1; j >= 0; j--) {
      const double lcur = lrow, rcur = rj;
      if (j > 0) { lrow = Lm[(size_t)(j - 1) * n + min(li, j - 1)]; rj = 1.0 / Lm[(size_t)(j - 1) * n + (j - 1)]; }   // next step's row, off the chain
      const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(bi) & 0xffffffffll), j);
      const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(bi) >> 32), j);
      const double xj = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo) * rcur;
      if (lane == j) bi = xj;
      else if (lane < j) bi -= lcur * xj;
    }
    if (lane < n) step[6 * slotCam[lane / 6] + lane % 6] = bi;
  } else {
    // the same chain through LDS for larger systems: wave-level fences instead of workgroup barriers
    __shared__ double bsh[128];
    double* b = bsh;
    for (int i = lane; i < n; i += 64) b[i] = y[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (int j = n - 1; j >= 0; j--) {
      const double xj = b[j] / Lm[(size_t)j * n + j];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      if (lane == 0) b[j] = xj;
      for (int i = lane; i < j; i += 64) b[i] -= Lm[(size_t)j * n + i] * xj;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < n; i += 64) step[6 * slotCam[i / 6] + i % 6] = b[i];
  }
}

// landmark steps by back-substitution (already negated), and each landmark's share of step.g and step^T H step.
// Four lanes per landmark (three of them carry one coordinate m each): the kernel is a latency chain of ~10 observations x 2 passes
// per landmark over 2000 landmarks, i.e. 8 workgroups with a thread per landmark (24 us); with a lane per coordinate it is 32
// workgroups and a third of the loads per lane.  The quad exchanges b and the step by DPP-width shuffles.
__global__ __launch_bounds__(256) void k_lm_backsub(int K, int L, const double* __restrict__ Hll, const double* __restrict__ g,
                                                    const int* __restrict__ lmStart, const int* __restrict__ lmObs,
                                                    const int* __restrict__ cam, const double* __restrict__ scale,
                                                    const unsigned char* __restrict__ active, const double* __restrict__ Vinv,
                                                    const double* __restrict__ Ws, double* __restrict__ step,
                                                    double* __restrict__ lmPart, LmStatus* __restrict__ st) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int l = t >> 2, m = t & 3;
  const bool on = l < L && m < 3;
  const int lc = min(l, L - 1), mc = min(m, 2);
  const int j0 = 6 * K + 3 * lc;
  const bool act = active[j0] != 0;
  const int e0 = lmStart[lc], e1 = lmStart[lc + 1];
  double bm = g[j0 + mc] * scale[j0 + mc];
  // A landmark's observations are a chain of dependent loads (slot -> camera -> 6 W entries and 6 step entries), twice.  For up to 16
  // observations the slots and cameras are fetched up front and the data in groups of four observations with all 48 loads in flight;
  // the sums keep their order.  More observations than that take the plain loops.
  const int nobs = e1 - e0;
  const bool quick = nobs <= 16;
  int pp[16], cc[16];
  if (act && quick) {
#pragma unroll
    for (int u = 0; u < 16; u++) pp[u] = u < nobs ? lmObs[e0 + u] : -1;
#pragma unroll
    for (int u = 0; u < 16; u++) cc[u] = pp[u] >= 0 ? cam[pp[u]] : 0;
#pragma unroll
    for (int u0 = 0; u0 < 16; u0 += 4) {
      if (u0 >= nobs) break;
      double wv[4][6], sv[4][6];
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (pp[u0 + u] >= 0) {
#pragma unroll
          for (int a = 0; a < 6; a++) { wv[u][a] = Ws[18 * (size_t)pp[u0 + u] + 3 * a + mc]; sv[u][a] = step[6 * cc[u0 + u] + a]; }
        }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (pp[u0 + u] >= 0) {
#pragma unroll
          for (int a = 0; a < 6; a++) bm -= wv[u][a] * sv[u][a];
        }
    }
  } else if (act) {
    for (int e = e0; e < e1; e++) {
      const int p = lmObs[e], c = cam[p];
      for (int a = 0; a < 6; a++) bm -= Ws[18 * (size_t)p + 3 * a + mc] * step[6 * c + a];
    }
  }
  const int base = (int)(threadIdx.x & 63) & ~3;
  const double b0 = __shfl(bm, base), b1 = __shfl(bm, base + 1), b2 = __shfl(bm, base + 2);
  const double* Vi = Vinv + 9 * (size_t)lc;
  const double slm = act ? -(Vi[3 * mc] * b0 + Vi[3 * mc + 1] * b1 + Vi[3 * mc + 2] * b2) : 0.0;
  if (on && act && !isfinite(slm)) st->finite = 0;
  if (on) step[j0 + m] = slm;
  const double s0 = __shfl(slm, base), s1 = __shfl(slm, base + 1), s2 = __shfl(slm, base + 2);
  // this lane's terms: coordinate m of step.g, row m of step^T H_ll step, column m of the cross terms 2 step_c^T W step_l
  double sg = 0.0, sHs = 0.0;
  if (act) {
    sg = slm * g[j0 + mc] * scale[j0 + mc];
    sHs = slm * scale[j0 + mc] * (Hll[9 * (size_t)lc + 3 * mc] * scale[j0] * s0 + Hll[9 * (size_t)lc + 3 * mc + 1] * scale[j0 + 1] * s1 +
                                  Hll[9 * (size_t)lc + 3 * mc + 2] * scale[j0 + 2] * s2);
    if (quick) {   // with the NEGATED camera step
#pragma unroll
      for (int u0 = 0; u0 < 16; u0 += 4) {
        if (u0 >= nobs) break;
        double wv[4][6], sv[4][6];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (pp[u0 + u] >= 0) {
#pragma unroll
            for (int a = 0; a < 6; a++) { wv[u][a] = Ws[18 * (size_t)pp[u0 + u] + 3 * a + mc]; sv[u][a] = step[6 * cc[u0 + u] + a]; }
          }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (pp[u0 + u] >= 0) {
            double w = 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) w += (-sv[u][a]) * wv[u][a];
            sHs += 2.0 * w * slm;
          }
      }
    } else {
      for (int e = e0; e < e1; e++) {
        const int p = lmObs[e], c = cam[p];
        double w = 0.0;
        for (int a = 0; a < 6; a++) w += (-step[6 * c + a]) * Ws[18 * (size_t)p + 3 * a + mc];
        sHs += 2.0 * w * slm;
      }
    }
  }
  if (m == 3) { sg = 0.0; sHs = 0.0; }
  // quad sums in a fixed order (m = 0, 1, 2)
  const double g0 = __shfl(sg, base), g1 = __shfl(sg, base + 1), g2 = __shfl(sg, base + 2);
  const double h0 = __shfl(sHs, base), h1 = __shfl(sHs, base + 1), h2 = __shfl(sHs, base + 2);
  if (l < L && m == 0) { lmPart[2 * (size_t)l] = (g0 + g1) + g2; lmPart[2 * (size_t)l + 1] = (h0 + h1) + h2; }
}

// candidate point x + Plus(step * scale) into the evaluation buffers; partial sums of |x - cand|^2 and |x|^2 per workgroup
__global__ __launch_bounds__(256) void k_lm_candidate(int K, int L, const double* __restrict__ q0, const double* __restrict__ t0,
                                                      const double* __restrict__ X0, const double* __restrict__ step,
                                                      const double* __restrict__ scale, const unsigned char* __restrict__ active,
                                                      double* __restrict__ q, double* __restrict__ t, double* __restrict__ X,
                                                      double* __restrict__ part) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sn = 0.0, xn = 0.0;
  if (i < K) {
    const int c = i;
    double cq[4], ct[3];
    if (active[6 * c]) {
      // the camera steps are stored as k_lm_chol solved them (not negated: k_lm_backsub reads them so); the landmarks' are negated
      const double d[3] = {-step[6 * c] * scale[6 * c], -step[6 * c + 1] * scale[6 * c + 1], -step[6 * c + 2] * scale[6 * c + 2]};
      quat_plus_dev(q0 + 4 * c, d, cq);
      for (int k = 0; k < 3; k++) ct[k] = t0[3 * c + k] + -step[6 * c + 3 + k] * scale[6 * c + 3 + k];
      for (int k = 0; k < 4; k++) { sn += (q0[4 * c + k] - cq[k]) * (q0[4 * c + k] - cq[k]); xn += q0[4 * c + k] * q0[4 * c + k]; }
      for (int k = 0; k < 3; k++) { sn += (t0[3 * c + k] - ct[k]) * (t0[3 * c + k] - ct[k]); xn += t0[3 * c + k] * t0[3 * c + k]; }
    } else {
      for (int k = 0; k < 4; k++) cq[k] = q0[4 * c + k];
      for (int k = 0; k < 3; k++) ct[k] = t0[3 * c + k];
    }
    for (int k = 0; k < 4; k++) q[4 * c + k] = cq[k];
    for (int k = 0; k < 3; k++) t[3 * c + k] = ct[k];
  } else if (i < K + L) {
    const int l = i - K, j0 = 6 * K + 3 * l;
    for (int k = 0; k < 3; k++) {
      double v = X0[3 * l + k];
      if (active[j0]) { const double c = v + step[j0 + k] * scale[j0 + k]; sn += (v - c) * (v - c); xn += v * v; v = c; }
      X[3 * l + k] = v;
    }
  }
  // both sums through one tree (block_sum_fixed's association for each)
  __shared__ double sm2[2][256];
  const int tid = threadIdx.x;
  sm2[0][tid] = sn; sm2[1][tid] = xn;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { sm2[0][tid] += sm2[0][tid + s]; sm2[1][tid] += sm2[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { part[2 * blockIdx.x] = sm2[0][0]; part[2 * blockIdx.x + 1] = sm2[1][0]; }
}

// publish the status record to the host's pinned copy (read after the stream synchronises: no D2H copy command), and — last kernel of
// a trial step — re-arm the device record for the next one (was a one-thread launch of its own)
// the record first, its sequence number after a system-scope fence: the host polls the number (dvs_ba_solve_device) instead of
// waiting for the stream
__device__ __forceinline__ void lm_publish(LmStatus* st, LmStatus* host) {
  const int seq = st->seq + 1;
  st->seq = seq;
  host->ok = st->ok; host->finite = st->finite; host->model_change = st->model_change; host->sn = st->sn; host->xn = st->xn;
  host->cand_cost = st->cand_cost; host->gmax = st->gmax; host->x_cost = st->x_cost; host->accept = st->accept;
  __threadfence_system();
  *reinterpret_cast<volatile int*>(&host->seq) = seq;
  __threadfence_system();
}

// Last kernel of a trial step: the step / point norms from k_lm_candidate's partial sums, the candidate's cost, and the model cost
// change -(step.g + step^T H step / 2) from the (negated) camera steps, the camera blocks and k_lm_backsub's landmark terms (was a
// launch of its own between back-substitution and candidate).
__global__ __launch_bounds__(256) void k_lm_norms(int nparts, const double* __restrict__ part, const double* __restrict__ cost, int K, int L,
                                                  const double* __restrict__ Hpp, const double* __restrict__ g,
                                                  const double* __restrict__ scale, const double* __restrict__ step,
                                                  const double* __restrict__ lmPart, double ptol, double ftol, LmStatus* __restrict__ st,
                                                  LmStatus* __restrict__ host) {
  const int tid = threadIdx.x;
  double sn = 0.0, xn = 0.0;
  for (int i = tid; i < nparts; i += 256) { sn += part[2 * i]; xn += part[2 * i + 1]; }
  double sg = 0.0, sHs = 0.0, fin = 0.0;
  for (int j = tid; j < 6 * K; j += 256) { sg += -step[j] * g[j] * scale[j]; if (!isfinite(step[j])) fin = 1.0; }
  for (int c = tid; c < K; c += 256)
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++)
      sHs += -step[6 * c + a] * scale[6 * c + a] * Hpp[36 * (size_t)c + 6 * a + b] * scale[6 * c + b] * -step[6 * c + b];
  for (int l = tid; l < L; l += 256) { sg += lmPart[2 * (size_t)l]; sHs += lmPart[2 * (size_t)l + 1]; }
  // the five sums through ONE tree (block_sum_fixed's association for each; five passes of it were 45 barriers)
  __shared__ double sm5[5][256];
  sm5[0][tid] = sn; sm5[1][tid] = xn; sm5[2][tid] = sg; sm5[3][tid] = sHs; sm5[4][tid] = fin;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 5; k++) sm5[k][tid] += sm5[k][tid + s];
    }
    __syncthreads();
  }
  const double a = sm5[0][0], b = sm5[1][0], tg = sm5[2][0], th = sm5[3][0], tf = sm5[4][0];
  if (tid == 0) {
    if (tf > 0.0) st->finite = 0;
    st->model_change = -(tg + 0.5 * th);
    st->sn = a; st->xn = b; st->cand_cost = *cost;
    // the verdict of dvs_ba_solve_device's loop, in its order and arithmetic (IEEE sqrt / divide on both sides): the launches that
    // follow an accepted step are already enqueued behind this kernel and read it
    int acc = 0;
    if (st->ok && st->finite && st->model_change > 0.0 && !(sqrt(a) <= ptol * (sqrt(b) + ptol))) {
      const double cost_change = st->x_cost - st->cand_cost;
      if (!(fabs(cost_change) <= ftol * st->x_cost)) acc = cost_change / st->model_change > 1e-3 ? 1 : 0;
    }
    st->accept = acc;
    lm_publish(st, host);
    st->ok = 1; st->finite = 1; st->model_change = 0; st->sn = 0; st->xn = 0; st->cand_cost = 0;
  }
}

// the accepted candidate becomes the point of the next iteration (one launch instead of three copy commands)
__global__ __launch_bounds__(256) void k_lm_accept(int K, int L, const double* __restrict__ q, const double* __restrict__ t,
                                                   const double* __restrict__ X, double* __restrict__ q0, double* __restrict__ t0,
                                                   double* __restrict__ X0, const int* __restrict__ gate) {
  if (gate && !*gate) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 4 * K) q0[i] = q[i];
  if (i < 3 * K) t0[i] = t[i];
  if (i < 3 * L) X0[i] = X[i];
}

// Ceres' gradient max-norm: |x - Plus(x, -g)|_inf over the active blocks, and the cost of the accepted point
__global__ __launch_bounds__(256) void k_lm_gmax(int K, int L, const double* __restrict__ q0, const double* __restrict__ g,
                                                 const unsigned char* __restrict__ active, const double* __restrict__ cost,
                                                 LmStatus* __restrict__ st, LmStatus* __restrict__ host, const int* __restrict__ gate) {
  if (gate && !*gate) return;
  __shared__ double sm[256];
  const int tid = threadIdx.x;
  double m = 0.0;
  for (int c = tid; c < K; c += 256) if (active[6 * c]) {
    const double d[3] = {-g[6 * c], -g[6 * c + 1], -g[6 * c + 2]};
    double qp[4];
    quat_plus_dev(q0 + 4 * c, d, qp);
    for (int i = 0; i < 4; i++) m = fmax(m, fabs(q0[4 * c + i] - qp[i]));
    for (int i = 0; i < 3; i++) m = fmax(m, fabs(g[6 * c + 3 + i]));
  }
  for (int l = tid; l < L; l += 256) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) m = fmax(m, fabs(g[6 * K + 3 * l + i]));
  // max is exact in any order: wavefront shuffles, then the four wavefronts' values (was an eight-barrier tree)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) sm[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) { st->gmax = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3])); st->x_cost = *cost; lm_publish(st, host); }
}

__global__ void k_lm_reset(LmStatus* st) { st->ok = 1; st->finite = 1; st->model_change = 0; st->sn = 0; st->xn = 0; st->cand_cost = 0; st->seq = 0; st->accept = 0; }

}  // namespace dvs

using namespace dvs;

struct dvs_ba {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  int K = 0, L = 0, R = 0, nChunks = 0, lmBlocks = 0;
  double fx = 0, fy = 0, cx = 0, cy = 0, sigma = 1, huber = 1.345;
  std::vector<double> q, t, X;            // host mirror of the parameters
  std::vector<int> cam, lm, perm;         // camera-sorted observation arrays, perm[p] = original index
  std::vector<int> lmStart, lmObs;
  std::vector<unsigned char> pose_fixed, lm_fixed;
  // Device memory of a problem is ONE grow-only arena (dvs_ba_set_problem) and of the LM workspace another (dvs_ba_solve_device); the
  // host-side tables reach the first through ONE pinned staging block and ONE asynchronous copy on the handle's stream.  A new window
  // of the same or a smaller shape (SlidingWindowBA::optimize, every 2 s in the reference) allocates nothing.
  struct Arena { uint8_t* base = nullptr; size_t cap = 0; } prob_arena, lm_arena;
  uint8_t* h_stage = nullptr; size_t h_stage_cap = 0;   // pinned
  bool lm_poll = true, lm_speculate = true;   // DVS_LM_POLL=0 / DVS_LM_SPECULATE=0 (read once in dvs_ba_create): A/B switches of dvs_ba_solve_device
  bool lm_ready = false;   // dvs_ba_solve_device: structure tables built and uploaded
  int lm_nc = 0;           // ... free cameras
  // device
  double *d_q = nullptr, *d_t = nullptr, *d_X = nullptr, *d_uv = nullptr;
  int *d_cam = nullptr, *d_lm = nullptr, *d_camChunkStart = nullptr, *d_lmStart = nullptr, *d_lmObs = nullptr;
  unsigned char *d_pf = nullptr, *d_lf = nullptr;
  BaChunk* d_chunks = nullptr;
  double *d_res = nullptr, *d_Jp = nullptr, *d_Jl = nullptr, *d_W = nullptr, *d_partial = nullptr;
  double *d_Hpp = nullptr, *d_Hll = nullptr, *d_g = nullptr, *d_cost = nullptr, *d_costCam = nullptr;
  int* d_ticket = nullptr;   // k_ba_reduce: arrival counter of its camera workgroups (the last one sums the cost)
  double *d_raw = nullptr;  // R*(2+8+6+6)
  // device LM (dvs_ba_solve_device): accepted point, scaling, LM diagonal, step, per-landmark inverses, scaled W, Y = W V^-1,
  // reduced system, observation-of-(landmark, camera) table
  double *d_q0 = nullptr, *d_t0 = nullptr, *d_X0 = nullptr, *d_scale = nullptr, *d_diag = nullptr, *d_step = nullptr, *d_Vinv = nullptr,
         *d_Ws = nullptr, *d_Y = nullptr, *d_S = nullptr, *d_rhs = nullptr, *d_lmPart = nullptr, *d_normPart = nullptr;
  int *d_obsOf = nullptr, *d_slotCam = nullptr;
  unsigned char* d_active = nullptr;
  dvs::LmStatus* d_status = nullptr;
  dvs::LmStatus* h_status = nullptr;  // pinned
  double* h_out = nullptr;            // pinned staging of the solved parameters (q, t, X)
  size_t h_out_cap = 0;
  const int* eval_gate = nullptr;     // see BaDev::gate
  bool eval_accept = false;           // see BaDev::acc_*
  std::vector<double> trace;          // dvs_ba_get_trace: 6 doubles per trust-region iteration of the last solve
  void log(double radius, int kind, double dc, double dm, double rel, double cand) {
    const double row[6] = {radius, (double)kind, dc, dm, rel, cand};
    trace.insert(trace.end(), row, row + 6);
  }
};

namespace {

// forget the current problem (the arenas and pinned blocks stay)
void ba_reset(dvs_ba* h) {
  if (h->d_raw) (void)hipFree(h->d_raw);
  h->d_q0 = h->d_t0 = h->d_X0 = h->d_scale = h->d_diag = h->d_step = h->d_Vinv = h->d_Ws = h->d_Y = h->d_S = h->d_rhs = h->d_lmPart = h->d_normPart = nullptr;
  h->d_obsOf = h->d_slotCam = nullptr; h->d_active = nullptr; h->d_status = nullptr;
  h->lm_ready = false;
  h->d_q = h->d_t = h->d_X = h->d_uv = nullptr; h->d_cam = h->d_lm = h->d_camChunkStart = h->d_lmStart = h->d_lmObs = nullptr;
  h->d_pf = h->d_lf = nullptr; h->d_chunks = nullptr; h->d_res = h->d_Jp = h->d_Jl = h->d_W = h->d_partial = nullptr;
  h->d_Hpp = h->d_Hll = h->d_g = h->d_cost = nullptr; h->d_raw = nullptr; h->d_costCam = nullptr; h->d_ticket = nullptr;
}

void ba_release(dvs_ba* h) {
  ba_reset(h);
  if (h->prob_arena.base) (void)hipFree(h->prob_arena.base);
  if (h->lm_arena.base) (void)hipFree(h->lm_arena.base);
  h->prob_arena = dvs_ba::Arena{}; h->lm_arena = dvs_ba::Arena{};
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  if (h->h_status) (void)hipHostFree(h->h_status);
  if (h->h_out) (void)hipHostFree(h->h_out);
  h->h_stage = nullptr; h->h_stage_cap = 0; h->h_status = nullptr; h->h_out = nullptr; h->h_out_cap = 0;
}

// layout of an arena: take() hands out 256-byte aligned offsets; bind() resolves them once the arena is large enough
struct ArenaPlan {
  size_t used = 0;
  size_t take(size_t bytes) { const size_t o = used; used += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return o; }
};
dvs_status arena_fit(dvs_ba::Arena& A, size_t need) {
  if (need <= A.cap) return DVS_OK;
  if (A.base) DVS_HIP(hipFree(A.base));
  A.base = nullptr; A.cap = 0;
  const size_t cap = need + need / 4;
  DVS_HIP(hipMalloc((void**)&A.base, cap));
  A.cap = cap;
  return DVS_OK;
}

BaDev dev_view(const dvs_ba* h) {
  BaDev P;
  P.q = h->d_q; P.t = h->d_t; P.X = h->d_X; P.uv = h->d_uv; P.cam = h->d_cam; P.lm = h->d_lm;
  P.pose_fixed = h->d_pf; P.lm_fixed = h->d_lf;
  P.fx = h->fx; P.fy = h->fy; P.cx = h->cx; P.cy = h->cy; P.inv_sigma = 1.0 / h->sigma; P.huber_a = h->huber;
  P.gate = h->eval_gate;
  P.acc_blocks = 0; P.acc_K = h->K; P.acc_L = h->L; P.acc_q0 = P.acc_t0 = P.acc_X0 = nullptr;
  if (h->eval_accept) { P.acc_blocks = std::max(1, (3 * h->L + 1023) / 1024); P.acc_q0 = h->d_q0; P.acc_t0 = h->d_t0; P.acc_X0 = h->d_X0; }
  return P;
}

// flags as k_ba_eval; withLm: also reduce landmark blocks
dvs_status enqueue_eval(dvs_ba* h, int flags, bool withLm) {
  if (h->R == 0) return DVS_OK;
  const BaDev P = dev_view(h);
  double* raw = h->d_raw;
  hipLaunchKernelGGL(k_ba_eval, dim3(h->nChunks + P.acc_blocks), dim3(256), 0, h->stream, P, h->d_chunks, flags, h->d_res, h->d_Jp, h->d_Jl, h->d_W,
                     h->d_partial, raw, raw ? raw + 2 * (size_t)h->R : nullptr, raw ? raw + 10 * (size_t)h->R : nullptr,
                     raw ? raw + 16 * (size_t)h->R : nullptr);
  // the reductions need every chunk's records: a second launch.  (Chained into the evaluation launch by arrival tickets they were
  // bit-identical and SLOWER — 17.2 vs 13.2 us for one window, 6.3 vs 1.33 us per window in a 64-window batch: an agent-scope release
  // per workgroup writes back the XCD's L2; profiles/r03_ba_chained_reduction_experiment.json.)
  hipLaunchKernelGGL(k_ba_reduce, dim3(h->lmBlocks + (h->K + 7) / 8), dim3(256), 0, h->stream, P, h->K, h->L, h->nChunks, h->d_chunks,
                     h->d_camChunkStart, h->d_lmStart, h->d_lmObs, h->d_res, h->d_Jl, h->d_partial, h->lmBlocks, withLm ? 1 : 0,
                     flags == 0 ? 1 : 0, h->d_Hpp, h->d_Hll, h->d_g, h->d_cost, h->d_costCam, h->d_ticket);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}

dvs_status upload_params(dvs_ba* h, const std::vector<double>& q, const std::vector<double>& t, const std::vector<double>& X) {
  DVS_HIP(hipMemcpyAsync(h->d_q, q.data(), q.size() * 8, hipMemcpyHostToDevice, h->stream));
  DVS_HIP(hipMemcpyAsync(h->d_t, t.data(), t.size() * 8, hipMemcpyHostToDevice, h->stream));
  DVS_HIP(hipMemcpyAsync(h->d_X, X.data(), X.size() * 8, hipMemcpyHostToDevice, h->stream));
  return DVS_OK;
}

// ceres::EigenQuaternionManifold::Plus on the raw (w,x,y,z) memory read as Eigen (x,y,z,w)
void quat_plus(const double* x, const double* d, double* o) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { for (int i = 0; i < 4; i++) o[i] = x[i]; return; }
  const double sn = sin(nd) / nd;
  const double dx = sn * d[0], dy = sn * d[1], dz = sn * d[2], dw = cos(nd);
  o[3] = dw * x[3] - dx * x[0] - dy * x[1] - dz * x[2];
  o[0] = dw * x[0] + dx * x[3] + dy * x[2] - dz * x[1];
  o[1] = dw * x[1] + dy * x[3] + dz * x[0] - dx * x[2];
  o[2] = dw * x[2] + dz * x[3] + dx * x[1] - dy * x[0];
}

bool chol_solve(std::vector<double>& A, int n, std::vector<double>& b) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0)) return false;
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
  return true;
}

bool inv3(const double* A, double* B) {
  const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], hh = A[7], i = A[8];
  const double det = a * (e * i - f * hh) - b * (d * i - f * g) + c * (d * hh - e * g);
  if (det == 0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  B[0] = (e * i - f * hh) * id; B[1] = (c * hh - b * i) * id; B[2] = (b * f - c * e) * id;
  B[3] = (f * g - d * i) * id; B[4] = (a * i - c * g) * id; B[5] = (c * d - a * f) * id;
  B[6] = (d * hh - e * g) * id; B[7] = (b * g - a * hh) * id; B[8] = (a * e - b * d) * id;
  return true;
}

}  // namespace

extern "C" {

dvs_status dvs_ba_create(int32_t device, dvs_ba** out) {
  DVS_ARG(out);
  *out = nullptr;
  DVS_TRY(check_device(device));
  dvs_ba* h = new (std::nothrow) dvs_ba();
  if (!h) { set_error("out of host memory"); return DVS_ERR_HIP; }
  h->device = device;
  hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; set_error("hipStreamCreate: %s", hipGetErrorString(e)); return DVS_ERR_HIP; }
  h->stream = h->own_stream;
  h->lm_poll = dvs::env_switch("DVS_LM_POLL", 1) != 0;
  h->lm_speculate = dvs::env_switch("DVS_LM_SPECULATE", 1) != 0;
  *out = h;
  return DVS_OK;
}

void dvs_ba_destroy(dvs_ba* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  ba_release(h);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

dvs_status dvs_ba_set_stream(dvs_ba* h, void* s) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)s;
  return DVS_OK;
}
dvs_status dvs_ba_synchronize(dvs_ba* h) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  return DVS_OK;
}

dvs_status dvs_ba_set_problem(dvs_ba* h, int32_t K, const double* q_wxyz, const double* t, int32_t L, const double* X, int32_t R,
                              const int32_t* cam_idx, const int32_t* lm_idx, const double* uv, const uint8_t* pose_fixed,
                              const uint8_t* lm_fixed, double fx, double fy, double cx, double cy, double sigma_pixels,
                              double huber_delta) {
  DVS_ARG(h && K >= 0 && L >= 0 && R >= 0);
  DVS_ARG((q_wxyz && t) || K == 0);
  DVS_ARG(X || L == 0);
  DVS_ARG((cam_idx && lm_idx && uv) || R == 0);
  for (int i = 0; i < R; i++) {
    if (cam_idx[i] < 0 || cam_idx[i] >= K || lm_idx[i] < 0 || lm_idx[i] >= L) { set_error("observation %d references camera %d / landmark %d out of range", i, cam_idx[i], lm_idx[i]); return DVS_ERR_ARG; }
  }
  DVS_HIP(hipSetDevice(h->device));
  DVS_HIP(hipStreamSynchronize(h->stream));
  ba_reset(h);
  h->K = K; h->L = L; h->R = R;
  h->fx = fx; h->fy = fy; h->cx = cx; h->cy = cy; h->sigma = sigma_pixels; h->huber = huber_delta;
  h->q.assign(q_wxyz, q_wxyz + 4 * (size_t)K); h->t.assign(t, t + 3 * (size_t)K); h->X.assign(X, X + 3 * (size_t)L);
  h->pose_fixed.assign(K, 0); h->lm_fixed.assign(L, 0);
  if (pose_fixed) h->pose_fixed.assign(pose_fixed, pose_fixed + K);
  if (lm_fixed) h->lm_fixed.assign(lm_fixed, lm_fixed + L);
  // observations grouped by camera (stable), chunks of <= 256 per workgroup
  std::vector<int> camCount(K + 1, 0);
  for (int i = 0; i < R; i++) camCount[cam_idx[i] + 1]++;
  for (int c = 0; c < K; c++) camCount[c + 1] += camCount[c];
  h->perm.assign(R, 0);
  {
    std::vector<int> cur(camCount.begin(), camCount.end() - 1);
    for (int i = 0; i < R; i++) h->perm[cur[cam_idx[i]]++] = i;
  }
  h->cam.resize(R); h->lm.resize(R);
  std::vector<double> uvp(2 * (size_t)R);
  for (int p = 0; p < R; p++) { const int i = h->perm[p]; h->cam[p] = cam_idx[i]; h->lm[p] = lm_idx[i]; uvp[2 * p] = uv[2 * i]; uvp[2 * p + 1] = uv[2 * i + 1]; }
  std::vector<BaChunk> chunks;
  std::vector<int> camChunkStart(K + 1, 0);
  for (int c = 0; c < K; c++) {
    camChunkStart[c] = (int)chunks.size();
    for (int s = camCount[c]; s < camCount[c + 1]; s += 256) chunks.push_back(BaChunk{c, s, std::min(256, camCount[c + 1] - s), 0});
  }
  camChunkStart[K] = (int)chunks.size();
  h->nChunks = (int)chunks.size();
  h->lmStart.assign(L + 1, 0);
  for (int p = 0; p < R; p++) h->lmStart[h->lm[p] + 1]++;
  for (int l = 0; l < L; l++) h->lmStart[l + 1] += h->lmStart[l];
  h->lmObs.assign(R, 0);
  {
    std::vector<int> cur(h->lmStart.begin(), h->lmStart.end() - 1);
    for (int p = 0; p < R; p++) h->lmObs[cur[h->lm[p]]++] = p;
  }
  h->lmBlocks = (L + 255) / 256;
  // one arena: [tables uploaded from the host | buffers that start at zero | evaluation outputs]
  ArenaPlan pl;
  const size_t Rz = std::max(R, 1), Kz = std::max(K, 1), Lz = std::max(L, 1), Cz = std::max(h->nChunks, 1);
  const size_t o_q = pl.take(Kz * 32), o_t = pl.take(Kz * 24), o_X = pl.take(Lz * 24), o_uv = pl.take(Rz * 16), o_cam = pl.take(Rz * 4), o_lm = pl.take(Rz * 4),
               o_ccs = pl.take((size_t)(K + 1) * 4), o_lms = pl.take((size_t)(L + 1) * 4), o_lmo = pl.take(Rz * 4), o_pf = pl.take(Kz), o_lf = pl.take(Lz),
               o_chunks = pl.take(Cz * sizeof(BaChunk));
  const size_t uploadBytes = pl.used;
  const size_t o_Hpp = pl.take(Kz * 36 * 8), o_Hll = pl.take(Lz * 9 * 8), o_g = pl.take((size_t)(6 * K + 3 * L + 1) * 8), o_cost = pl.take(8), o_costCam = pl.take(Kz * 8),
               o_ticket = pl.take(4);
  const size_t zeroBytes = pl.used - uploadBytes;
  const size_t o_res = pl.take(Rz * 2 * 8), o_Jp = pl.take(Rz * 12 * 8), o_Jl = pl.take(Rz * 6 * 8), o_W = pl.take(Rz * 18 * 8), o_partial = pl.take(Cz * 28 * 8);
  DVS_TRY(arena_fit(h->prob_arena, pl.used));
  if (uploadBytes > h->h_stage_cap) {
    if (h->h_stage) DVS_HIP(hipHostFree(h->h_stage));
    h->h_stage = nullptr; h->h_stage_cap = 0;
    DVS_HIP(hipHostMalloc((void**)&h->h_stage, uploadBytes + uploadBytes / 4));
    h->h_stage_cap = uploadBytes + uploadBytes / 4;
  }
  uint8_t* B = h->prob_arena.base; uint8_t* S = h->h_stage;
  auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) memcpy(S + off, src, bytes); };
  put(o_q, h->q.data(), h->q.size() * 8); put(o_t, h->t.data(), h->t.size() * 8); put(o_X, h->X.data(), h->X.size() * 8);
  put(o_uv, uvp.data(), uvp.size() * 8); put(o_cam, h->cam.data(), h->cam.size() * 4); put(o_lm, h->lm.data(), h->lm.size() * 4);
  put(o_ccs, camChunkStart.data(), camChunkStart.size() * 4); put(o_lms, h->lmStart.data(), h->lmStart.size() * 4); put(o_lmo, h->lmObs.data(), h->lmObs.size() * 4);
  put(o_pf, h->pose_fixed.data(), h->pose_fixed.size()); put(o_lf, h->lm_fixed.data(), h->lm_fixed.size()); put(o_chunks, chunks.data(), chunks.size() * sizeof(BaChunk));
  h->d_q = (double*)(B + o_q); h->d_t = (double*)(B + o_t); h->d_X = (double*)(B + o_X); h->d_uv = (double*)(B + o_uv);
  h->d_cam = (int*)(B + o_cam); h->d_lm = (int*)(B + o_lm); h->d_camChunkStart = (int*)(B + o_ccs); h->d_lmStart = (int*)(B + o_lms); h->d_lmObs = (int*)(B + o_lmo);
  h->d_pf = B + o_pf; h->d_lf = B + o_lf; h->d_chunks = (BaChunk*)(B + o_chunks);
  h->d_Hpp = (double*)(B + o_Hpp); h->d_Hll = (double*)(B + o_Hll); h->d_g = (double*)(B + o_g); h->d_cost = (double*)(B + o_cost);
  h->d_costCam = (double*)(B + o_costCam); h->d_ticket = (int*)(B + o_ticket);
  h->d_res = (double*)(B + o_res); h->d_Jp = (double*)(B + o_Jp); h->d_Jl = (double*)(B + o_Jl); h->d_W = (double*)(B + o_W); h->d_partial = (double*)(B + o_partial);
  // one copy, one fill, both on the handle's stream: whatever the caller enqueues next on it is ordered behind them
  DVS_HIP(hipMemcpyAsync(B, S, uploadBytes, hipMemcpyHostToDevice, h->stream));
  DVS_HIP(hipMemsetAsync(B + uploadBytes, 0, zeroBytes, h->stream));
  return DVS_OK;
}

dvs_status dvs_ba_evaluate(dvs_ba* h, double* cost, double* residuals, double* J_pose, double* J_lm, double* grad) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_TRY(enqueue_eval(h, 1, true));
  DVS_HIP(hipStreamSynchronize(h->stream));
  const int R = h->R;
  if (cost) { *cost = 0; if (R) DVS_HIP(hipMemcpy(cost, h->d_cost, 8, hipMemcpyDeviceToHost)); }
  auto unpermute = [&](double* dst, const double* dsrc, int width) -> dvs_status {
    std::vector<double> tmp((size_t)R * width);
    if (R) DVS_HIP(hipMemcpy(tmp.data(), dsrc, tmp.size() * 8, hipMemcpyDeviceToHost));
    for (int p = 0; p < R; p++) memcpy(dst + (size_t)h->perm[p] * width, &tmp[(size_t)p * width], (size_t)width * 8);
    return DVS_OK;
  };
  if (residuals) DVS_TRY(unpermute(residuals, h->d_res, 2));
  if (J_pose) DVS_TRY(unpermute(J_pose, h->d_Jp, 12));
  if (J_lm) DVS_TRY(unpermute(J_lm, h->d_Jl, 6));
  if (grad) {
    memset(grad, 0, (size_t)(6 * h->K + 3 * h->L) * 8);
    if (R) DVS_HIP(hipMemcpy(grad, h->d_g, (size_t)(6 * h->K + 3 * h->L) * 8, hipMemcpyDeviceToHost));
  }
  return DVS_OK;
}

dvs_status dvs_ba_evaluate_raw(dvs_ba* h, double* residuals, double* J_q, double* J_t, double* J_X) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  const int R = h->R;
  if (R == 0) return DVS_OK;
  if (!h->d_raw) DVS_HIP(hipMalloc((void**)&h->d_raw, (size_t)R * 22 * 8));
  DVS_TRY(enqueue_eval(h, 4, false));
  DVS_HIP(hipStreamSynchronize(h->stream));
  std::vector<double> tmp((size_t)R * 22);
  DVS_HIP(hipMemcpy(tmp.data(), h->d_raw, tmp.size() * 8, hipMemcpyDeviceToHost));
  for (int p = 0; p < R; p++) {
    const size_t i = h->perm[p];
    if (residuals) memcpy(residuals + 2 * i, &tmp[2 * (size_t)p], 16);
    if (J_q) memcpy(J_q + 8 * i, &tmp[2 * (size_t)R + 8 * (size_t)p], 64);
    if (J_t) memcpy(J_t + 6 * i, &tmp[10 * (size_t)R + 6 * (size_t)p], 48);
    if (J_X) memcpy(J_X + 6 * i, &tmp[16 * (size_t)R + 6 * (size_t)p], 48);
  }
  return DVS_OK;
}

dvs_status dvs_ba_normal_equations(dvs_ba* h, double* H_pp, double* H_ll, double* W, double* g, double* cost) {
  DVS_ARG(h);
  DVS_HIP(hipSetDevice(h->device));
  DVS_TRY(enqueue_eval(h, 1 | 2, true));
  DVS_HIP(hipStreamSynchronize(h->stream));
  const int R = h->R, K = h->K, L = h->L;
  if (H_pp) DVS_HIP(hipMemcpy(H_pp, h->d_Hpp, (size_t)K * 36 * 8, hipMemcpyDeviceToHost));
  if (H_ll) DVS_HIP(hipMemcpy(H_ll, h->d_Hll, (size_t)L * 9 * 8, hipMemcpyDeviceToHost));
  if (g) DVS_HIP(hipMemcpy(g, h->d_g, (size_t)(6 * K + 3 * L) * 8, hipMemcpyDeviceToHost));
  if (cost) DVS_HIP(hipMemcpy(cost, h->d_cost, 8, hipMemcpyDeviceToHost));
  if (W && R) {
    std::vector<double> tmp((size_t)R * 18);
    DVS_HIP(hipMemcpy(tmp.data(), h->d_W, tmp.size() * 8, hipMemcpyDeviceToHost));
    for (int p = 0; p < R; p++) memcpy(W + 18 * (size_t)h->perm[p], &tmp[18 * (size_t)p], 144);
  }
  return DVS_OK;
}

dvs_status dvs_ba_evaluate_device(dvs_ba* h, int32_t iters) {
  DVS_ARG(h && iters >= 0);
  DVS_HIP(hipSetDevice(h->device));
  for (int i = 0; i < iters; i++) DVS_TRY(enqueue_eval(h, 1, true));
  return DVS_OK;
}

dvs_status dvs_ba_get_trace(const dvs_ba* h, double* rows, int32_t cap_rows, int32_t* n_rows) {
  DVS_ARG(h && n_rows && cap_rows >= 0);
  const int n = (int)(h->trace.size() / 6);
  *n_rows = n;
  if (rows) memcpy(rows, h->trace.data(), (size_t)std::min(n, cap_rows) * 6 * sizeof(double));
  return DVS_OK;
}

dvs_status dvs_ba_get_parameters(dvs_ba* h, double* q_wxyz, double* t, double* X) {
  DVS_ARG(h);
  if (q_wxyz) memcpy(q_wxyz, h->q.data(), h->q.size() * 8);
  if (t) memcpy(t, h->t.data(), h->t.size() * 8);
  if (X) memcpy(X, h->X.data(), h->X.size() * 8);
  return DVS_OK;
}

// Levenberg-Marquardt with ceres::Solver defaults (trust_region_minimizer.cc / levenberg_marquardt_strategy.cc, Ceres 2.x):
// initial radius 1e4, Jacobi column scaling fixed at the first Jacobian, LM diagonal clamped to [1e-6, 1e32], step
// accepted when relative decrease > 1e-3, radius /= max(1/3, 1 - (2 rho - 1)^3) on success, /= 2,4,8.. on failure;
// parameter / function tolerance tested on the candidate BEFORE acceptance; landmarks eliminated by a Schur complement.
dvs_status dvs_ba_solve(dvs_ba* h, int32_t max_iterations, double ftol, double gtol, double ptol, dvs_ba_summary* summary) {
  DVS_ARG(h && summary && max_iterations >= 0);
  memset(summary, 0, sizeof(*summary));
  summary->termination = 2;
  summary->linear_solver = 2;   // reduced camera system + Cholesky on the HOST (this entry point)
  DVS_HIP(hipSetDevice(h->device));
  const int K = h->K, L = h->L, R = h->R, NT = 6 * K + 3 * L;
  if (R == 0) { set_error("no observations"); return DVS_ERR_ARG; }
  std::vector<double> Hpp((size_t)K * 36), Hll((size_t)L * 9), W((size_t)R * 18), g(NT);
  double x_cost = 0;
  auto evaluate_full = [&]() -> dvs_status {
    DVS_TRY(enqueue_eval(h, 1 | 2, true));
    DVS_HIP(hipMemcpyAsync(Hpp.data(), h->d_Hpp, Hpp.size() * 8, hipMemcpyDeviceToHost, h->stream));
    DVS_HIP(hipMemcpyAsync(Hll.data(), h->d_Hll, Hll.size() * 8, hipMemcpyDeviceToHost, h->stream));
    DVS_HIP(hipMemcpyAsync(W.data(), h->d_W, W.size() * 8, hipMemcpyDeviceToHost, h->stream));  // camera-sorted order
    DVS_HIP(hipMemcpyAsync(g.data(), h->d_g, g.size() * 8, hipMemcpyDeviceToHost, h->stream));
    DVS_HIP(hipMemcpyAsync(&x_cost, h->d_cost, 8, hipMemcpyDeviceToHost, h->stream));
    DVS_HIP(hipStreamSynchronize(h->stream));
    return DVS_OK;
  };
  h->trace.clear();
  std::vector<double> q = h->q, t = h->t, X = h->X;
  DVS_TRY(upload_params(h, q, t, X));
  DVS_TRY(evaluate_full());
  summary->initial_cost = x_cost;
  double min_cost = x_cost;

  std::vector<unsigned char> lmUsed(L, 0), camUsed(K, 0), active(NT, 0);
  for (int p = 0; p < R; p++) { lmUsed[h->lm[p]] = 1; camUsed[h->cam[p]] = 1; }
  std::vector<int> camSlot(K, -1);
  int nc = 0;
  for (int c = 0; c < K; c++) if (!h->pose_fixed[c] && camUsed[c]) { camSlot[c] = nc++; for (int a = 0; a < 6; a++) active[6 * c + a] = 1; }
  for (int l = 0; l < L; l++) if (!h->lm_fixed[l] && lmUsed[l]) for (int a = 0; a < 3; a++) active[6 * K + 3 * l + a] = 1;
  const int n = 6 * nc;
  std::vector<double> scale(NT, 1.0), diagonal(NT, 0.0), step(NT, 0.0), Vinv((size_t)L * 9), Sm, rhs;
  for (int c = 0; c < K; c++) for (int a = 0; a < 6; a++) scale[6 * c + a] = 1.0 / (1.0 + sqrt(Hpp[36 * (size_t)c + 7 * a]));
  for (int l = 0; l < L; l++) for (int a = 0; a < 3; a++) scale[6 * K + 3 * l + a] = 1.0 / (1.0 + sqrt(Hll[9 * (size_t)l + 4 * a]));

  auto grad_max_norm = [&]() {
    double m = 0;
    for (int c = 0; c < K; c++) if (camSlot[c] >= 0) {
      const double d[3] = {-g[6 * c], -g[6 * c + 1], -g[6 * c + 2]};
      double qp[4];
      quat_plus(&q[4 * c], d, qp);
      for (int i = 0; i < 4; i++) m = std::max(m, fabs(q[4 * c + i] - qp[i]));
      for (int i = 0; i < 3; i++) m = std::max(m, fabs(g[6 * c + 3 + i]));
    }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) m = std::max(m, fabs(g[6 * K + 3 * l + i]));
    return m;
  };
  auto x_norm = [&]() {
    double s = 0;
    for (int c = 0; c < K; c++) if (camSlot[c] >= 0) { for (int i = 0; i < 4; i++) s += q[4 * c + i] * q[4 * c + i]; for (int i = 0; i < 3; i++) s += t[3 * c + i] * t[3 * c + i]; }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) s += X[3 * l + i] * X[3 * l + i];
    return sqrt(s);
  };

  double radius = 1e4, decrease_factor = 2.0, gmax = grad_max_norm();
  bool reuse_diagonal = false;
  int iteration = 0, invalid = 0;
  summary->termination = 1;
  while (true) {
    if (iteration >= max_iterations) { summary->termination = 1; break; }
    if (gmax <= gtol) { summary->termination = 0; break; }
    if (radius < 1e-32) { summary->termination = 0; break; }
    iteration++;
    if (!reuse_diagonal) {
      for (int c = 0; c < K; c++) for (int a = 0; a < 6; a++) { const int j = 6 * c + a; diagonal[j] = std::min(std::max(Hpp[36 * (size_t)c + 7 * a] * scale[j] * scale[j], 1e-6), 1e32); }
      for (int l = 0; l < L; l++) for (int a = 0; a < 3; a++) { const int j = 6 * K + 3 * l + a; diagonal[j] = std::min(std::max(Hll[9 * (size_t)l + 4 * a] * scale[j] * scale[j], 1e-6), 1e32); }
    }
    reuse_diagonal = true;
    // reduced camera system S = (H_pp + D) - sum_l Wl (H_ll + D)^-1 Wl^T, on the Jacobi-scaled blocks
    Sm.assign((size_t)n * n, 0.0); rhs.assign(n, 0.0);
    for (int c = 0; c < K; c++) if (camSlot[c] >= 0) {
      const int o = 6 * camSlot[c];
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) Sm[(size_t)(o + a) * n + o + b] = Hpp[36 * (size_t)c + 6 * a + b] * scale[6 * c + a] * scale[6 * c + b];
        Sm[(size_t)(o + a) * n + o + a] += diagonal[6 * c + a] / radius;
        rhs[o + a] = g[6 * c + a] * scale[6 * c + a];
      }
    }
    bool ok = true;
    double Ws[16][18], Y[16][18];
    std::vector<double> Wbig, Ybig;
    for (int l = 0; l < L && ok; l++) {
      const int j0 = 6 * K + 3 * l;
      if (!active[j0]) continue;
      double V[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) V[3 * a + b] = Hll[9 * (size_t)l + 3 * a + b] * scale[j0 + a] * scale[j0 + b];
      for (int a = 0; a < 3; a++) V[4 * a] += diagonal[j0 + a] / radius;
      double* Vi = &Vinv[9 * (size_t)l];
      if (!inv3(V, Vi)) { ok = false; break; }
      const double gl[3] = {g[j0] * scale[j0], g[j0 + 1] * scale[j0 + 1], g[j0 + 2] * scale[j0 + 2]};
      const int e0 = h->lmStart[l], ne = h->lmStart[l + 1] - e0;
      double (*Wl)[18] = Ws; double (*Yl)[18] = Y;
      if (ne > 16) { Wbig.resize((size_t)ne * 18); Ybig.resize((size_t)ne * 18); Wl = (double(*)[18])Wbig.data(); Yl = (double(*)[18])Ybig.data(); }
      for (int e = 0; e < ne; e++) {
        const int p = h->lmObs[e0 + e], c = h->cam[p];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) Wl[e][3 * a + b] = W[18 * (size_t)p + 3 * a + b] * scale[6 * c + a] * scale[j0 + b];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) Yl[e][3 * a + b] = Wl[e][3 * a] * Vi[b] + Wl[e][3 * a + 1] * Vi[3 + b] + Wl[e][3 * a + 2] * Vi[6 + b];
      }
      for (int e = 0; e < ne; e++) {
        const int ci = camSlot[h->cam[h->lmObs[e0 + e]]];
        if (ci < 0) continue;
        for (int a = 0; a < 6; a++) rhs[6 * ci + a] -= Yl[e][3 * a] * gl[0] + Yl[e][3 * a + 1] * gl[1] + Yl[e][3 * a + 2] * gl[2];
        for (int f = 0; f < ne; f++) {
          const int ck = camSlot[h->cam[h->lmObs[e0 + f]]];
          if (ck < 0) continue;
          double* dst = &Sm[(size_t)(6 * ci) * n + 6 * ck];
          for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++)
            dst[(size_t)a * n + b] -= Yl[e][3 * a] * Wl[f][3 * b] + Yl[e][3 * a + 1] * Wl[f][3 * b + 1] + Yl[e][3 * a + 2] * Wl[f][3 * b + 2];
        }
      }
    }
    if (ok && n > 0) ok = chol_solve(Sm, n, rhs);
    bool valid = ok;
    double model_cost_change = 0;
    if (ok) {
      std::fill(step.begin(), step.end(), 0.0);
      for (int c = 0; c < K; c++) if (camSlot[c] >= 0) for (int a = 0; a < 6; a++) step[6 * c + a] = rhs[6 * camSlot[c] + a];
      for (int l = 0; l < L; l++) {
        const int j0 = 6 * K + 3 * l;
        if (!active[j0]) continue;
        double b[3] = {g[j0] * scale[j0], g[j0 + 1] * scale[j0 + 1], g[j0 + 2] * scale[j0 + 2]};
        for (int e = h->lmStart[l]; e < h->lmStart[l + 1]; e++) {
          const int p = h->lmObs[e], c = h->cam[p];
          if (camSlot[c] < 0) continue;
          for (int m = 0; m < 3; m++) for (int a = 0; a < 6; a++) b[m] -= W[18 * (size_t)p + 3 * a + m] * scale[6 * c + a] * scale[j0 + m] * step[6 * c + a];
        }
        const double* Vi = &Vinv[9 * (size_t)l];
        for (int a = 0; a < 3; a++) step[j0 + a] = Vi[3 * a] * b[0] + Vi[3 * a + 1] * b[1] + Vi[3 * a + 2] * b[2];
      }
      for (int j = 0; j < NT; j++) { step[j] = -step[j]; if (!std::isfinite(step[j])) valid = false; }
      if (valid) {
        double sg = 0, sHs = 0;
        for (int j = 0; j < NT; j++) sg += step[j] * g[j] * scale[j];
        for (int c = 0; c < K; c++) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++)
          sHs += step[6 * c + a] * scale[6 * c + a] * Hpp[36 * (size_t)c + 6 * a + b] * scale[6 * c + b] * step[6 * c + b];
        for (int l = 0; l < L; l++) { const int j0 = 6 * K + 3 * l; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sHs += step[j0 + a] * scale[j0 + a] * Hll[9 * (size_t)l + 3 * a + b] * scale[j0 + b] * step[j0 + b]; }
        for (int p = 0; p < R; p++) { const int c = h->cam[p], j0 = 6 * K + 3 * h->lm[p]; for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) sHs += 2.0 * step[6 * c + a] * scale[6 * c + a] * W[18 * (size_t)p + 3 * a + b] * scale[j0 + b] * step[j0 + b]; }
        model_cost_change = -(sg + 0.5 * sHs);
        if (model_cost_change <= 0.0) valid = false;
      }
    }
    if (!valid) {
      h->log(radius, 0, 0, model_cost_change, 0, 0);
      if (++invalid >= 5) { summary->termination = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = false;
      continue;
    }
    invalid = 0;
    std::vector<double> cq = q, ct = t, cX = X;
    for (int c = 0; c < K; c++) if (camSlot[c] >= 0) {
      const double d[3] = {step[6 * c] * scale[6 * c], step[6 * c + 1] * scale[6 * c + 1], step[6 * c + 2] * scale[6 * c + 2]};
      quat_plus(&q[4 * c], d, &cq[4 * c]);
      for (int i = 0; i < 3; i++) ct[3 * c + i] = t[3 * c + i] + step[6 * c + 3 + i] * scale[6 * c + 3 + i];
    }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) cX[3 * l + i] = X[3 * l + i] + step[6 * K + 3 * l + i] * scale[6 * K + 3 * l + i];
    DVS_TRY(upload_params(h, cq, ct, cX));
    DVS_TRY(enqueue_eval(h, 0, false));  // cost only
    double cand_cost = 0;
    DVS_HIP(hipMemcpyAsync(&cand_cost, h->d_cost, 8, hipMemcpyDeviceToHost, h->stream));
    DVS_HIP(hipStreamSynchronize(h->stream));
    double sn = 0;
    for (int c = 0; c < K; c++) if (camSlot[c] >= 0) { for (int i = 0; i < 4; i++) sn += (q[4 * c + i] - cq[4 * c + i]) * (q[4 * c + i] - cq[4 * c + i]); for (int i = 0; i < 3; i++) sn += (t[3 * c + i] - ct[3 * c + i]) * (t[3 * c + i] - ct[3 * c + i]); }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) sn += (X[3 * l + i] - cX[3 * l + i]) * (X[3 * l + i] - cX[3 * l + i]);
    if (sqrt(sn) <= ptol * (x_norm() + ptol)) { h->log(radius, 3, x_cost - cand_cost, model_cost_change, 0, cand_cost); summary->termination = 0; break; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= ftol * x_cost) { h->log(radius, 4, cost_change, model_cost_change, 0, cand_cost); summary->termination = 0; break; }
    const double rel = cost_change / model_cost_change;
    h->log(radius, rel > 1e-3 ? 1 : 2, cost_change, model_cost_change, rel, cand_cost);
    if (rel > 1e-3) {
      q = cq; t = ct; X = cX;
      DVS_TRY(evaluate_full());  // parameters on the device already are the candidate
      gmax = grad_max_norm();
      summary->num_successful_steps++;
      min_cost = std::min(min_cost, x_cost);
      radius = radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  summary->num_iterations = iteration;
  summary->final_cost = min_cost;
  h->q = q; h->t = t; h->X = X;
  DVS_TRY(upload_params(h, q, t, X));
  DVS_HIP(hipStreamSynchronize(h->stream));
  return DVS_OK;
}

// The same trust-region loop as dvs_ba_solve with every O(R) step on the device (kernels above): per trial step the host
// enqueues system -> Cholesky -> back-substitution -> candidate -> cost evaluation, reads ONE status record, decides.
dvs_status dvs_ba_solve_device(dvs_ba* h, int32_t max_iterations, double ftol, double gtol, double ptol, dvs_ba_summary* summary) {
  DVS_ARG(h && summary && max_iterations >= 0);
  memset(summary, 0, sizeof(*summary));
  summary->termination = 2;
  summary->linear_solver = 1;   // linear algebra on the DEVICE
  DVS_HIP(hipSetDevice(h->device));
  const int K = h->K, L = h->L, R = h->R, NT = 6 * K + 3 * L;
  if (R == 0) { set_error("no observations"); return DVS_ERR_ARG; }
  // active blocks, camera slots, observation table: fixed for the life of the handle (the observation structure and the fixed flags
  // are set at creation), built and uploaded by the first solve
  if (!h->lm_ready) {
    std::vector<unsigned char> lmUsed(L, 0), camUsed(K, 0), active(NT, 0);
    for (int p = 0; p < R; p++) { lmUsed[h->lm[p]] = 1; camUsed[h->cam[p]] = 1; }
    std::vector<int> slotCam;
    for (int c = 0; c < K; c++) if (!h->pose_fixed[c] && camUsed[c]) { slotCam.push_back(c); for (int a = 0; a < 6; a++) active[6 * c + a] = 1; }
    for (int l = 0; l < L; l++) if (!h->lm_fixed[l] && lmUsed[l]) for (int a = 0; a < 3; a++) active[6 * K + 3 * l + a] = 1;
    const int nc = (int)slotCam.size();
    if (K > 64 || nc > 16 || nc == 0) {
      set_error("dvs_ba_solve_device handles sliding windows (<= 64 cameras, 1..16 of them free); this problem has %d / %d", K, nc);
      return DVS_ERR_UNSUPPORTED;
    }
    std::vector<int> obsOf((size_t)L * K, -1);
    for (int l = 0; l < L; l++)
      for (int e = h->lmStart[l]; e < h->lmStart[l + 1]; e++) {
        const int p = h->lmObs[e];
        int& slot = obsOf[(size_t)l * K + h->cam[p]];
        if (slot >= 0) { set_error("landmark %d is observed twice in camera %d: use dvs_ba_solve", l, h->cam[p]); return DVS_ERR_UNSUPPORTED; }
        slot = p;
      }
    hipStream_t st = h->stream;
    {
      ArenaPlan pl;
      const size_t Rz = std::max(R, 1), Kz = std::max(K, 1), Lz = std::max(L, 1);
      const size_t o_obsOf = pl.take(obsOf.size() * 4 + 4), o_slot = pl.take(64 * 4), o_active = pl.take((size_t)NT + 1);
      const size_t uploadBytes = pl.used;
      const size_t o_q0 = pl.take(Kz * 32), o_t0 = pl.take(Kz * 24), o_X0 = pl.take(Lz * 24), o_scale = pl.take((size_t)NT * 8), o_diag = pl.take((size_t)NT * 8),
                   o_step = pl.take((size_t)NT * 8), o_Vinv = pl.take(Lz * 72), o_Ws = pl.take(Rz * 144), o_Y = pl.take(Rz * 144),
                   o_S = pl.take((size_t)(1 + kSchurSplit) * 96 * 96 * 8), o_rhs = pl.take((size_t)(1 + kSchurSplit) * 96 * 8), o_lmPart = pl.take(Lz * 16),
                   o_normPart = pl.take((size_t)((K + L + 255) / 256 + 1) * 16), o_status = pl.take(sizeof(LmStatus));
      DVS_TRY(arena_fit(h->lm_arena, pl.used));
      if (uploadBytes > h->h_stage_cap) {
        DVS_HIP(hipStreamSynchronize(st));   // an upload of dvs_ba_set_problem may still be reading the block
        if (h->h_stage) DVS_HIP(hipHostFree(h->h_stage));
        h->h_stage = nullptr; h->h_stage_cap = 0;
        DVS_HIP(hipHostMalloc((void**)&h->h_stage, uploadBytes + uploadBytes / 4));
        h->h_stage_cap = uploadBytes + uploadBytes / 4;
      }
      uint8_t* B = h->lm_arena.base;
      h->d_obsOf = (int*)(B + o_obsOf); h->d_slotCam = (int*)(B + o_slot); h->d_active = B + o_active;
      h->d_q0 = (double*)(B + o_q0); h->d_t0 = (double*)(B + o_t0); h->d_X0 = (double*)(B + o_X0); h->d_scale = (double*)(B + o_scale);
      h->d_diag = (double*)(B + o_diag); h->d_step = (double*)(B + o_step); h->d_Vinv = (double*)(B + o_Vinv); h->d_Ws = (double*)(B + o_Ws); h->d_Y = (double*)(B + o_Y);
      h->d_S = (double*)(B + o_S); h->d_rhs = (double*)(B + o_rhs); h->d_lmPart = (double*)(B + o_lmPart); h->d_normPart = (double*)(B + o_normPart);
      h->d_status = (LmStatus*)(B + o_status);
      if (!h->h_status) {
        DVS_HIP(hipHostMalloc((void**)&h->h_status, 2 * sizeof(LmStatus)));   // [0]: the trial's record (k_lm_norms), [1]: the point's (k_lm_gmax)
        DVS_HIP(hipFuncSetAttribute((const void*)k_lm_chol, hipFuncAttributeMaxDynamicSharedMemorySize, 97 * 96 * 8));
      }
      const size_t outBytes = ((size_t)7 * Kz + 3 * Lz) * 8;
      if (outBytes > h->h_out_cap) {
        if (h->h_out) DVS_HIP(hipHostFree(h->h_out));
        h->h_out = nullptr; h->h_out_cap = 0;
        DVS_HIP(hipHostMalloc((void**)&h->h_out, outBytes + outBytes / 4));
        h->h_out_cap = outBytes + outBytes / 4;
      }
      DVS_HIP(hipStreamSynchronize(st));     // the staging block is free (dvs_ba_set_problem's upload has completed)
      memcpy(h->h_stage + o_obsOf, obsOf.data(), obsOf.size() * 4);
      memcpy(h->h_stage + o_slot, slotCam.data(), (size_t)nc * 4);
      memcpy(h->h_stage + o_active, active.data(), (size_t)NT);
      DVS_HIP(hipMemcpyAsync(B, h->h_stage, uploadBytes, hipMemcpyHostToDevice, st));
    }
    h->lm_nc = nc;
    h->lm_ready = true;
  }
  const int nc = h->lm_nc, n = 6 * nc;
  h->trace.clear();
  hipStream_t st = h->stream;
  DVS_TRY(upload_params(h, h->q, h->t, h->X));
  const dim3 copyGrid((std::max(4 * K, 3 * L) + 255) / 256);   // one launch instead of three copy commands
  hipLaunchKernelGGL(k_lm_accept, copyGrid, dim3(256), 0, st, K, L, h->d_q, h->d_t, h->d_X, h->d_q0, h->d_t0, h->d_X0, nullptr);
  // two host records: the gated k_lm_gmax of an accepted step runs while the host may still be reading the trial's numbers, so it
  // must not publish over them
  LmStatus* S = h->h_status;
  LmStatus* SP = h->h_status + 1;
  // the last kernel enqueued wrote the record into the pinned host copy (lm_publish): poll its sequence number — a bounded spin,
  // then the stream wait — instead of sleeping in hipStreamSynchronize (a wake-up per trial step and per accepted step)
  int expect_seq = 0;
  S->seq = 0; SP->seq = 0;
  const bool poll = h->lm_poll;
  auto fetch_status = [&](const LmStatus* rec) -> dvs_status {
    expect_seq++;
    if (poll) {
      const volatile int* seq = &rec->seq;
      const auto t0 = std::chrono::steady_clock::now();
      for (int spin = 1; *seq != expect_seq; spin++) {
        __builtin_ia32_pause();
        if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      if (*seq == expect_seq) return DVS_OK;
    }
    DVS_HIP(hipStreamSynchronize(st));
    return DVS_OK;
  };
  // Jacobian blocks, gradient, cost of the point in the evaluation buffers; `gate`: only if the device's verdict says so
  // accept: the evaluation buffers hold a candidate that becomes the point of the next iteration (stored by extra workgroups of
  // the evaluation kernel)
  auto enqueue_full = [&](const int* gate, bool accept) -> dvs_status {
    h->eval_gate = gate; h->eval_accept = accept;
    const dvs_status e = enqueue_eval(h, 1 | 2, true);
    h->eval_gate = nullptr; h->eval_accept = false;
    DVS_TRY(e);
    hipLaunchKernelGGL(k_lm_gmax, dim3(1), dim3(256), 0, st, K, L, h->d_q0, h->d_g, h->d_active, h->d_cost, h->d_status, SP, gate);
    return DVS_OK;
  };
  auto evaluate_full = [&](bool accept) -> dvs_status { DVS_TRY(enqueue_full(nullptr, accept)); return fetch_status(SP); };
  // The launches that follow an accepted step (accept, full evaluation, gradient norm: ~27 us of host launch time) are enqueued right
  // behind the trial, gated on the verdict k_lm_norms leaves in the status record, so that they are ready when the trial ends; the
  // host takes the same decision from the same numbers.  Should the two ever differ (a last-bit difference between the host's and the
  // device's sqrt / pow in a tolerance test), the DEVICE's verdict stands — it has already been applied to the buffers.
  const int* verdict = &h->d_status->accept;
  const bool speculate = h->lm_speculate;
  hipLaunchKernelGGL(k_lm_reset, dim3(1), dim3(1), 0, st, h->d_status);
  DVS_TRY(evaluate_full(false));
  double x_cost = SP->x_cost, gmax = SP->gmax;
  summary->initial_cost = x_cost;
  double min_cost = x_cost;
  hipLaunchKernelGGL(k_lm_scale, dim3((NT + 255) / 256), dim3(256), 0, st, K, L, h->d_Hpp, h->d_Hll, h->d_scale);

  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int iteration = 0, invalid = 0;
  const int nparts = (K + L + 255) / 256;
  summary->termination = 1;
  while (true) {
    if (iteration >= max_iterations) { summary->termination = 1; break; }
    if (gmax <= gtol) { summary->termination = 0; break; }
    if (radius < 1e-32) { summary->termination = 0; break; }
    iteration++;
    hipLaunchKernelGGL(k_lm_observations, dim3((R + L + K + 255) / 256), dim3(256), 0, st, K, L, R, h->d_Hpp, h->d_Hll, h->d_W, h->d_cam, h->d_lm,
                       h->d_lmStart, h->d_lmObs, h->d_scale, h->d_diag, h->d_active, radius, reuse_diagonal ? 0 : 1, h->d_Vinv, h->d_Ws, h->d_Y,
                       h->d_status);
    reuse_diagonal = true;
    hipLaunchKernelGGL(k_lm_schur, dim3(nc, nc, kSchurSplit), dim3(256), 0, st, K, L, n, h->d_slotCam, h->d_obsOf, h->d_active, h->d_Hpp, h->d_g, h->d_scale,
                       h->d_diag, radius, h->d_Ws, h->d_Y, h->d_S, h->d_rhs);
    hipLaunchKernelGGL(k_lm_chol, dim3(1), dim3(kCholThreads), (size_t)(n + 1) * n * 8, st, K, n, h->d_slotCam, h->d_S, h->d_rhs, h->d_step, h->d_status);
    hipLaunchKernelGGL(k_lm_backsub, dim3((4 * L + 255) / 256), dim3(256), 0, st, K, L, h->d_Hll, h->d_g, h->d_lmStart, h->d_lmObs, h->d_cam,
                       h->d_scale, h->d_active, h->d_Vinv, h->d_Ws, h->d_step, h->d_lmPart, h->d_status);
    hipLaunchKernelGGL(k_lm_candidate, dim3(nparts), dim3(256), 0, st, K, L, h->d_q0, h->d_t0, h->d_X0, h->d_step, h->d_scale, h->d_active,
                       h->d_q, h->d_t, h->d_X, h->d_normPart);
    DVS_TRY(enqueue_eval(h, 0, false));  // cost of the candidate
    hipLaunchKernelGGL(k_lm_norms, dim3(1), dim3(256), 0, st, nparts, h->d_normPart, h->d_cost, K, L, h->d_Hpp, h->d_g, h->d_scale, h->d_step,
                       h->d_lmPart, ptol, ftol, h->d_status, S);
    if (speculate) DVS_TRY(enqueue_full(verdict, true));
    DVS_HIP(hipGetLastError());
    DVS_TRY(fetch_status(S));
    const bool valid = S->ok && S->finite && S->model_change > 0.0;
    const bool dev_accept = speculate && S->accept != 0;   // the gated launches ran: the candidate IS the point of the next iteration
    if (!valid && !dev_accept) {
      h->log(radius, 0, 0, S->model_change, 0, 0);
      if (++invalid >= 5) { summary->termination = 2; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = false;
      continue;
    }
    invalid = 0;
    const double cost_change = x_cost - S->cand_cost;
    if (!dev_accept) {
      if (sqrt(S->sn) <= ptol * (sqrt(S->xn) + ptol)) { h->log(radius, 3, cost_change, S->model_change, 0, S->cand_cost); summary->termination = 0; break; }
      if (fabs(cost_change) <= ftol * x_cost) { h->log(radius, 4, cost_change, S->model_change, 0, S->cand_cost); summary->termination = 0; break; }
    }
    const double rel = cost_change / S->model_change;
    const bool accept = speculate ? dev_accept : rel > 1e-3;
    h->log(radius, accept ? 1 : 2, cost_change, S->model_change, rel, S->cand_cost);
    if (accept) {
      if (speculate) {
        DVS_TRY(fetch_status(SP));       // the gated launches ran: wait for k_lm_gmax's record
      } else {
        DVS_TRY(evaluate_full(true));
      }
      x_cost = SP->x_cost; gmax = SP->gmax;
      summary->num_successful_steps++;
      min_cost = std::min(min_cost, x_cost);
      radius = radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  summary->num_iterations = iteration;
  summary->final_cost = min_cost;
  // the accepted point becomes the problem's parameters (host mirror and evaluation buffers)
  hipLaunchKernelGGL(k_lm_accept, copyGrid, dim3(256), 0, st, K, L, h->d_q0, h->d_t0, h->d_X0, h->d_q, h->d_t, h->d_X, nullptr);
  // ... the host mirror through the handle's pinned block, written by a kernel: the three device-to-host copy commands this replaces
  // now and then blocked for 7 ms when enqueued (first solve after a warm-up, pageable or pinned destination alike)
  double* ho = h->h_out;
  hipLaunchKernelGGL(k_lm_accept, copyGrid, dim3(256), 0, st, K, L, h->d_q0, h->d_t0, h->d_X0, ho, ho + 4 * (size_t)K, ho + 7 * (size_t)K, nullptr);
  DVS_HIP(hipStreamSynchronize(st));
  memcpy(h->q.data(), ho, (size_t)K * 32); memcpy(h->t.data(), ho + 4 * (size_t)K, (size_t)K * 24);
  memcpy(h->X.data(), ho + 7 * (size_t)K, (size_t)L * 24);
  return DVS_OK;
}

// CameraPose::fromRt / toRt (bundle_adjustment.hpp:138-165, 192-212) with Eigen 3.4's Quaterniond(Matrix3d),
// normalize() and toRotationMatrix() arithmetic.  R is row-major 3x3, poses in the caller's convention (the backend
// passes camera-to-world; fromRt inverts).  Pure host arithmetic: part of the drop-in adapter, not of the hot path.
dvs_status dvs_ba_pose_from_rt(const double* R_wc, const double* t_wc, double* q_wxyz, double* trans) {
  DVS_ARG(R_wc && t_wc && q_wxyz && trans);
  double m[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = R_wc[3 * j + i];  // R_camera_world = R^T
  for (int i = 0; i < 3; i++) trans[i] = -(m[i][0] * t_wc[0] + m[i][1] * t_wc[1] + m[i][2] * t_wc[2]);
  double qv[4];  // x, y, z, w
  double tr = m[0][0] + m[1][1] + m[2][2];
  if (tr > 0) {
    tr = sqrt(tr + 1.0);
    qv[3] = 0.5 * tr;
    tr = 0.5 / tr;
    qv[0] = (m[2][1] - m[1][2]) * tr; qv[1] = (m[0][2] - m[2][0]) * tr; qv[2] = (m[1][0] - m[0][1]) * tr;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    tr = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    qv[i] = 0.5 * tr;
    tr = 0.5 / tr;
    qv[3] = (m[k][j] - m[j][k]) * tr;
    qv[j] = (m[j][i] + m[i][j]) * tr;
    qv[k] = (m[k][i] + m[i][k]) * tr;
  }
  const double nn = sqrt(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2] + qv[3] * qv[3]);
  q_wxyz[0] = qv[3] / nn; q_wxyz[1] = qv[0] / nn; q_wxyz[2] = qv[1] / nn; q_wxyz[3] = qv[2] / nn;
  return DVS_OK;
}

dvs_status dvs_ba_pose_to_rt(const double* q_wxyz, const double* trans, double* R_wc, double* t_wc) {
  DVS_ARG(q_wxyz && trans && R_wc && t_wc);
  const double w = q_wxyz[0], x = q_wxyz[1], y = q_wxyz[2], z = q_wxyz[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double Rcw[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_wc[3 * i + j] = Rcw[j][i];
  for (int i = 0; i < 3; i++) t_wc[i] = -(R_wc[3 * i] * trans[0] + R_wc[3 * i + 1] * trans[1] + R_wc[3 * i + 2] * trans[2]);
  return DVS_OK;
}

}  // extern "C"
