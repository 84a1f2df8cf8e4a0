// ransac.hip — the frontend's two robust-estimation stages as BATCHED HYPOTHESIS kernels:
//   dvs_find_fundamental_ransac   cv::findFundamentalMat(p1, p2, mask, FM_RANSAC, 2.0, 0.99)     frontend.cpp:635, 1146-1147
//   dvs_solve_pnp_ransac          cv::solvePnPRansac(obj, img, K, dist, rvec, tvec, false, 100, 4.0, 0.99, inliers)
//                                                                                                frontend.cpp:911-921
// OpenCV runs these as sequential sample -> fit -> count loops on one CPU thread.  Here ALL hypotheses of a call are fitted
// at once (one thread each), scored at once (one workgroup per hypothesis over all correspondences), and a single thread then
// replays the sequential loop over the counts — RANSACUpdateNumIters's adaptive stopping rule included — so the result is what
// the sequential loop with the same samples would return.
//
// Two forms of the fundamental-matrix stage: dvs_find_fundamental_cv follows OpenCV's own procedure (cv::RNG sample sequence, 7-point
// solver: see k_f7_hypotheses below); dvs_find_fundamental_ransac and the PnP stage are the library's own estimators — the reference
// only consumes the inlier SET (frontend.cpp:640-644, 1149-1153) and the refined pose — over a sampler stated here and in DESIGN.md: sample j of hypothesis h draws r = splitmix64(seed + 0x9E3779B97F4A7C15
// * (h * 16 + j + 1)) mod (n - j) and takes the r-th index not drawn before (ascending), i.e. a uniform draw without
// replacement; parity is a tolerance on the inlier set and the pose (tests/test_gpu_ransac.py), not bit equality.
// Minimal solvers: normalised 8-point (null vector by complete-pivoting elimination, rank 2 enforced through the smallest
// right singular vector) where OpenCV's kernel is the 7-point one; P3P (Grunert's quartic, all <= 4 poses scored) where
// OpenCV's kernel is 5-point EPnP.  The final pose is refined on the inliers by Levenberg-Marquardt on the reprojection error,
// which is what SOLVEPNP_ITERATIVE does, so it does not depend on the minimal solver.
// All arithmetic is FP64 (a few hundred hypotheses of a few dozen flops: latency-bound, not a throughput kernel).
#include <float.h>
#include <math.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "common.h"
#ifdef DVS_TEST_HOOKS
#include "../../include/dvslam_hip_test.h"
#endif
#include "io_pinned.h"

namespace dvs {

dvs_status matcher_scratch(dvs_matcher* m, int slot, size_t bytes, void** out);
dvs_status matcher_pinned(dvs_matcher* m, size_t bytes, void** out, int** h_seq, int** counter);
hipStream_t matcher_stream(dvs_matcher* m);
int matcher_device(dvs_matcher* m);

__host__ __device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// One RANSAC problem of a batch (blockIdx.y of every kernel below): its correspondences are rows [off, off + n) of the point arrays,
// its hypotheses / counts / results slot `b` of the per-problem arrays.  The single-problem entry points are batches of one.
struct RansacProb { int off, n; unsigned long long seed; };

// k distinct indices out of n (k <= 8), uniform without replacement, in draw order
template <int KS>
__device__ __forceinline__ void sample_distinct(unsigned long long seed, int h, int n, int* idx) {
  int sorted[KS];
#pragma unroll
  for (int j = 0; j < KS; j++) {
    int r = (int)(splitmix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(h * 16 + j + 1)) % (unsigned long long)(n - j));
    for (int i = 0; i < j; i++) if (r >= sorted[i]) r++;   // skip the indices drawn before (ascending)
    idx[j] = r;
    int p = j;
    while (p > 0 && sorted[p - 1] > r) { sorted[p] = sorted[p - 1]; p--; }
    sorted[p] = r;
  }
}

// eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi)
__device__ __forceinline__ void smallest_eigvec3(const double* S, double* v) {
  double a[3][3] = {{S[0], S[1], S[2]}, {S[1], S[4], S[5]}, {S[2], S[5], S[8]}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; sweep++) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  int m = 0;
  if (a[1][1] < a[m][m]) m = 1;
  if (a[2][2] < a[m][m]) m = 2;
  for (int k = 0; k < 3; k++) v[k] = V[k][m];
}

// ---------------------------------------------------------------------------------------------------------------------------
// fundamental matrix: one thread = one hypothesis from 8 correspondences (x2^T F x1 = 0)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_f_hypotheses(const float* __restrict__ p1, const float* __restrict__ p2, const RansacProb* __restrict__ probs, int H,
                                                     double* __restrict__ Fout, int* __restrict__ valid) {
  const int h = blockIdx.x * 64 + threadIdx.x;
  if (h >= H) return;
  const RansacProb pb = probs[blockIdx.y];
  const int n = pb.n;
  p1 += 2 * (size_t)pb.off; p2 += 2 * (size_t)pb.off;
  Fout += 9 * (size_t)H * blockIdx.y; valid += (size_t)H * blockIdx.y;
  if (n < 8) { valid[h] = 0; return; }
  int idx[8];
  sample_distinct<8>(pb.seed, h, n, idx);
  double x1[8], y1[8], x2[8], y2[8];
  double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
  for (int i = 0; i < 8; i++) {
    x1[i] = p1[2 * idx[i]]; y1[i] = p1[2 * idx[i] + 1]; x2[i] = p2[2 * idx[i]]; y2[i] = p2[2 * idx[i] + 1];
    c1x += x1[i]; c1y += y1[i]; c2x += x2[i]; c2y += y2[i];
  }
  c1x /= 8; c1y /= 8; c2x /= 8; c2y /= 8;
  double d1 = 0, d2 = 0;
  for (int i = 0; i < 8; i++) {
    d1 += sqrt((x1[i] - c1x) * (x1[i] - c1x) + (y1[i] - c1y) * (y1[i] - c1y));
    d2 += sqrt((x2[i] - c2x) * (x2[i] - c2x) + (y2[i] - c2y) * (y2[i] - c2y));
  }
  bool ok = d1 > 1e-9 && d2 > 1e-9;
  const double s1 = ok ? sqrt(2.0) * 8 / d1 : 1.0, s2 = ok ? sqrt(2.0) * 8 / d2 : 1.0;   // Hartley: mean distance sqrt(2)
  double A[8][9];
  for (int i = 0; i < 8; i++) {
    const double u1 = (x1[i] - c1x) * s1, v1 = (y1[i] - c1y) * s1, u2 = (x2[i] - c2x) * s2, v2 = (y2[i] - c2y) * s2;
    A[i][0] = u2 * u1; A[i][1] = u2 * v1; A[i][2] = u2; A[i][3] = v2 * u1; A[i][4] = v2 * v1; A[i][5] = v2; A[i][6] = u1; A[i][7] = v1; A[i][8] = 1.0;
  }
  // null vector of the 8 x 9 system by Gauss-Jordan elimination with complete pivoting; the column never chosen is the free one.
  // Every loop is unrolled and the run-time row / column (pivot position) is applied by selects, so that the 72 matrix entries stay
  // in registers: indexed with run-time subscripts the array lived in scratch memory and this kernel took 102 us.  Same operations
  // in the same order on the same values.
  int colOf[8];
  unsigned usedMask = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (ok) {
      int pr = k, pc = -1;
      double best = 0;
#pragma unroll
      for (int i = k; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const double a = fabs(A[i][j]);
          const bool take = !((usedMask >> j) & 1u) && a > best;
          best = take ? a : best; pr = take ? i : pr; pc = take ? j : pc;
        }
      if (pc < 0 || best < 1e-12) {
        ok = false;
      } else {
        // rows k and pr change places
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const double tmp = A[k][j];
          double apr = tmp;
#pragma unroll
          for (int i = k + 1; i < 8; i++) apr = i == pr ? A[i][j] : apr;
          A[k][j] = apr;
#pragma unroll
          for (int i = k + 1; i < 8; i++) A[i][j] = i == pr ? tmp : A[i][j];
        }
        usedMask |= 1u << pc; colOf[k] = pc;
        double piv = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) piv = j == pc ? A[k][j] : piv;
        const double inv = 1.0 / piv;
#pragma unroll
        for (int j = 0; j < 9; j++) A[k][j] *= inv;
#pragma unroll
        for (int i = 0; i < 8; i++)
          if (i != k) {
            double fcoef = 0;
#pragma unroll
            for (int j = 0; j < 9; j++) fcoef = j == pc ? A[i][j] : fcoef;
            const bool nz = fcoef != 0.0;
#pragma unroll
            for (int j = 0; j < 9; j++) A[i][j] = nz ? A[i][j] - fcoef * A[k][j] : A[i][j];
          }
      }
    }
  }
  double f[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) {
    int fc = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) if (!((usedMask >> j) & 1u)) fc = j;
#pragma unroll
    for (int j = 0; j < 9; j++) f[j] = j == fc ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      double afc = 0;
#pragma unroll
      for (int j = 0; j < 9; j++) afc = j == fc ? A[k][j] : afc;
#pragma unroll
      for (int j = 0; j < 9; j++) f[j] = j == colOf[k] ? -afc : f[j];
    }
    // rank 2: F <- F - (F v) v^T with v the right singular vector of the smallest singular value
    double S[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) S[3 * a + b] = f[a] * f[b] + f[3 + a] * f[3 + b] + f[6 + a] * f[6 + b];
    double v[3];
    smallest_eigvec3(S, v);
    for (int r = 0; r < 3; r++) {
      const double fv = f[3 * r] * v[0] + f[3 * r + 1] * v[1] + f[3 * r + 2] * v[2];
      for (int cidx = 0; cidx < 3; cidx++) f[3 * r + cidx] -= fv * v[cidx];
    }
    // denormalise: F = T2^T Fn T1, T = [s 0 -s cx; 0 s -s cy; 0 0 1]
    double G[9];
    // Fn T1
    for (int r = 0; r < 3; r++) {
      G[3 * r] = f[3 * r] * s1; G[3 * r + 1] = f[3 * r + 1] * s1;
      G[3 * r + 2] = f[3 * r + 2] - s1 * (f[3 * r] * c1x + f[3 * r + 1] * c1y);
    }
    // T2^T (.)
    for (int cidx = 0; cidx < 3; cidx++) {
      f[cidx] = s2 * G[cidx]; f[3 + cidx] = s2 * G[3 + cidx];
      f[6 + cidx] = G[6 + cidx] - s2 * (c2x * G[cidx] + c2y * G[3 + cidx]);
    }
    double nrm = 0;
    for (int k = 0; k < 9; k++) nrm += f[k] * f[k];
    ok = nrm > 0 && isfinite(nrm);
    if (ok) { nrm = 1.0 / sqrt(nrm); for (int k = 0; k < 9; k++) f[k] *= nrm; }
  }
  for (int k = 0; k < 9; k++) Fout[9 * (size_t)h + k] = ok ? f[k] : 0.0;
  valid[h] = ok ? 1 : 0;
}

// OpenCV's FMEstimatorCallback::computeError: max of the two squared point-to-epipolar-line distances
__device__ __forceinline__ double epi_err(const double* F, double x1, double y1, double x2, double y2) {
  const double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  const double d2 = x2 * a + y2 * b + c, s2 = 1.0 / (a * a + b * b);
  const double a1 = F[0] * x2 + F[3] * y2 + F[6], b1 = F[1] * x2 + F[4] * y2 + F[7], c1 = F[2] * x2 + F[5] * y2 + F[8];
  const double d1 = x1 * a1 + y1 * b1 + c1, s1 = 1.0 / (a1 * a1 + b1 * b1);
  return fmax(d1 * d1 * s1, d2 * d2 * s2);
}

__device__ __forceinline__ int block_count256(bool pred) {  // number of threads of the 256-thread block with pred
  __shared__ int wcnt[4];
  const unsigned long long b = __ballot(pred);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  return wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// one workgroup per hypothesis: inlier count over all correspondences
__global__ __launch_bounds__(256) void k_f_score(const float* __restrict__ p1, const float* __restrict__ p2, const RansacProb* __restrict__ probs, int H,
                                                 const double* __restrict__ Fall, const int* __restrict__ valid, double thr2, int* __restrict__ counts, int fcmp) {
  const int h = blockIdx.x;
  const RansacProb pb = probs[blockIdx.y];
  const int n = pb.n;
  p1 += 2 * (size_t)pb.off; p2 += 2 * (size_t)pb.off;
  Fall += 9 * (size_t)H * blockIdx.y; valid += (size_t)H * blockIdx.y; counts += (size_t)H * blockIdx.y;
  if (!valid[h]) { if (threadIdx.x == 0) counts[h] = 0; return; }
  double F[9];
  for (int k = 0; k < 9; k++) F[k] = Fall[9 * (size_t)h + k];
  int total = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    bool in = false;
    if (i < n) {
      const double e = epi_err(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
      in = fcmp ? (float)e <= (float)thr2 : e <= thr2;   // OpenCV stores the error and the squared threshold as floats (findInliers)
    }
    total += block_count256(in);
  }
  if (threadIdx.x == 0) counts[h] = total;
}

// cv::RANSACUpdateNumIters
__device__ __forceinline__ int ransac_update_iters(double p, double ep, int modelPoints, int maxIters) {
  p = fmax(p, 0.0); p = fmin(p, 1.0);
  ep = fmax(ep, 0.0); ep = fmin(ep, 1.0);
  double num = fmax(1.0 - p, DBL_MIN);
  double denom = 1.0 - pow(1.0 - ep, (double)modelPoints);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

static int ransac_update_iters_host(double p, double ep, int modelPoints, int maxIters) {   // the same on the host (LMedS' iteration count)
  p = std::max(p, 0.0); p = std::min(p, 1.0);
  ep = std::max(ep, 0.0); ep = std::min(ep, 1.0);
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, (double)modelPoints);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)std::rint(num / denom);
}

// the sequential RANSAC loop replayed over the hypothesis counts (RANSACPointSetRegistrator::run): hypothesis h is iteration h,
// a strictly better count replaces the best and shortens the loop.  sel[0] = best hypothesis (-1: none), sel[1] = iterations used.
__global__ void k_ransac_select(const int* __restrict__ counts, int H, const RansacProb* __restrict__ probs, int modelPoints, double confidence, int group,
                                int* __restrict__ sel, int capInSeed = 0, int maxItersTrue = 0) {
  if (threadIdx.x != 0) return;
  const int n = probs[blockIdx.x].n;
  counts += (size_t)H * blockIdx.x; sel += 4 * (size_t)blockIdx.x;
  // cv mode: only the first H / group iterations of a loop of up to maxItersTrue have their models here (the loop usually stops long
  // before); the seed field carries how many iterations getSubset found a sample for.  sel[3] = 1: the loop wanted to go on past them
  const int avail = H / group;
  int niters = maxItersTrue > 0 ? maxItersTrue : avail, best = -1, bestCount = 0, it = 0;
  const int maxIters = niters;
  const int found = capInSeed ? (int)probs[blockIdx.x].seed : avail;
  for (; it < niters && it < avail && it < found; it++) {
    for (int s = 0; s < group; s++) {   // `group` candidate models per iteration (P3P: up to 4 poses per sample), in order
      const int h = it * group + s;
      const int good = counts[h];
      if (good > max(bestCount, modelPoints - 1)) {
        bestCount = good; best = h;
        niters = ransac_update_iters(confidence, (double)(n - good) / n, modelPoints, maxIters);
      }
    }
  }
  sel[0] = best; sel[1] = it; sel[2] = bestCount;
  sel[3] = it < niters && it >= avail && found >= avail ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_f_mask(const float* __restrict__ p1, const float* __restrict__ p2, const RansacProb* __restrict__ probs, int H,
                                                const double* __restrict__ Fall, const int* __restrict__ sel, double thr2, unsigned char* __restrict__ mask,
                                                double* __restrict__ Fbest, int fcmp) {
  const RansacProb pb = probs[blockIdx.y];
  const int n = pb.n;
  p1 += 2 * (size_t)pb.off; p2 += 2 * (size_t)pb.off; mask += pb.off;
  Fall += 9 * (size_t)H * blockIdx.y; sel += 4 * (size_t)blockIdx.y; Fbest += 9 * (size_t)blockIdx.y;
  const int best = sel[0];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (best < 0) { if (i < n) mask[i] = 0; if (i < 9) Fbest[i] = 0.0; return; }
  double F[9];
  for (int k = 0; k < 9; k++) F[k] = Fall[9 * (size_t)best + k];
  if (i < 9) Fbest[i] = F[i];
  if (i < n) {
    const double e = epi_err(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
    mask[i] = (fcmp ? (float)e <= (float)thr2 : e <= thr2) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// cv::findFundamentalMat(FM_RANSAC) as OpenCV 4.x runs it for >= 15 points (calib3d fundam.cpp / ptsetreg.cpp, restated from the
// published algorithm — the library is not in this image, PARITY UNPINNED):
//   * RANSACPointSetRegistrator(modelPoints = 7): one cv::RNG, seeded with (uint64)-1, threaded through all iterations; getSubset
//     draws index i as rng.uniform(0, n) = next() % n, drawn again while it repeats an earlier index of the sample; a complete
//     sample whose LAST point is collinear with two earlier ones in either image (FMEstimatorCallback::checkSubset ->
//     haveCollinearPoints) is drawn again as a whole, up to 10000 attempts;
//   * cv::RNG::next(): multiply-with-carry, state = (uint32)state * 4164903690 + (state >> 32), value = (uint32)state;
//   * run7Point: the two null vectors f1, f2 of the 7 x 9 system of RAW pixel coordinates, det(lambda f1 + (1 - lambda) f2) = 0 by
//     cv::solveCubic (closed form), one model per real root, scaled to F(3,3) = 1; every model of a sample is scored in order;
//   * findInliers: error and squared threshold compared as floats; a strictly better count replaces the best model and shortens the
//     loop (RANSACUpdateNumIters with modelPoints = 7).  No refit on the inliers.
// The sample sequence is sequential by nature (a repeated index or a rejected sample shifts everything behind it): it is made on
// the HOST (cv_subsets: microseconds) and uploaded with the points; the models of all samples are fitted and scored at once as in
// the library's own estimator above, and the sequential loop is replayed over the counts.  OpenCV takes the null space from an SVD
// (this kernel: complete-pivoting elimination): the same models to rounding, but the order of a sample's models follows the basis,
// so a tie in inlier count between two models of ONE sample may resolve differently.
// ---------------------------------------------------------------------------------------------------------------------------
struct CvRng {
  unsigned long long state;
  explicit CvRng(unsigned long long s = 0xffffffffull) : state(s ? s : 0xffffffffull) {}
  unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

static bool cv_have_collinear(const float* pts, const int* idx, int count) {   // the LAST point against every earlier pair
  const int i = count - 1;
  for (int j = 0; j < i; j++) {
    const double dx1 = pts[2 * idx[j]] - pts[2 * idx[i]], dy1 = pts[2 * idx[j] + 1] - pts[2 * idx[i] + 1];
    for (int k = 0; k < j; k++) {
      const double dx2 = pts[2 * idx[k]] - pts[2 * idx[i]], dy2 = pts[2 * idx[k] + 1] - pts[2 * idx[i] + 1];
      if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
    }
  }
  return false;
}

// the samples of iterations 0 .. iters - 1 (modelPoints indices each); returns how many iterations found one
static int cv_subsets(const float* p1, const float* p2, int n, int modelPoints, int iters, int32_t* idx_out, int maxAttempts = 10000) {
  CvRng rng(~0ull);
  int idx[16];
  for (int it = 0; it < iters; it++) {
    bool found = false;
    for (int attempt = 0; attempt < maxAttempts && !found; attempt++) {
      for (int i = 0; i < modelPoints; i++) {
        int v;
        bool dup;
        do {
          v = rng.uniform(0, n);
          dup = false;
          for (int j = 0; j < i; j++) dup = dup || idx[j] == v;
        } while (dup);
        idx[i] = v;
      }
      found = !(cv_have_collinear(p1, idx, modelPoints) || cv_have_collinear(p2, idx, modelPoints));
    }
    if (!found) return it;
    for (int i = 0; i < modelPoints; i++) idx_out[(size_t)it * modelPoints + i] = idx[i];
  }
  return iters;
}

// cv::solveCubic for c[0] x^3 + c[1] x^2 + c[2] x + c[3] (closed form, roots in the order OpenCV returns them)
__device__ __forceinline__ int cv_solve_cubic(const double* c, double* x) {
  double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
  int n = 0;
  x[0] = x[1] = x[2] = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0) return 0;
      x[0] = -a3 / a2; return 1;
    }
    double d = a2 * a2 - 4 * a1 * a3;
    if (d >= 0) {
      d = sqrt(d);
      const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
      if (fabs(q1) > fabs(q2)) { x[0] = q1 / a1; x[1] = a3 / q1; } else { x[0] = q2 / a1; x[1] = a3 / q2; }
      n = d > 0 ? 2 : 1;
    }
    return n;
  }
  a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
  const double Q = (a1 * a1 - 3 * a2) * (1. / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
  const double Qcubed = Q * Q * Q;
  double d = Qcubed - R * R;
  if (d > 0) {
    const double theta = acos(R / sqrt(Qcubed)), sqrtQ = sqrt(Q);
    const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
    x[0] = t0 * cos(t1) - t2;
    x[1] = t0 * cos(t1 + (2. * 3.1415926535897932384626433832795 / 3)) - t2;
    x[2] = t0 * cos(t1 + (4. * 3.1415926535897932384626433832795 / 3)) - t2;
    n = 3;
  } else if (d == 0) {
    if (R >= 0) { x[0] = -2 * pow(R, 1. / 3) - a1 / 3; x[1] = pow(R, 1. / 3) - a1 / 3; }
    else { x[0] = 2 * pow(-R, 1. / 3) - a1 / 3; x[1] = -pow(-R, 1. / 3) - a1 / 3; }
    n = x[0] == x[1] ? 1 : 2;
    if (n == 1) x[1] = 0;
  } else {
    d = sqrt(-d);
    double e = pow(d + fabs(R), 1. / 3);
    if (R > 0) e = -e;
    x[0] = (e + Q / e) - a1 * (1. / 3);
    n = 1;
  }
  return n;
}

__device__ __forceinline__ double det3rows(const double* a, const double* b, const double* c) {   // rows a, b, c
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
}

// one thread = one sample: up to three models at Fout[3 h .. 3 h + 2]
__global__ __launch_bounds__(64) void k_f7_hypotheses(const float* __restrict__ p1, const float* __restrict__ p2, const RansacProb* __restrict__ probs,
                                                      const int32_t* __restrict__ samples, int H, double* __restrict__ Fout, int* __restrict__ valid) {
  const int h = blockIdx.x * 64 + threadIdx.x;
  if (h >= H) return;
  const RansacProb pb = probs[blockIdx.y];
  p1 += 2 * (size_t)pb.off; p2 += 2 * (size_t)pb.off;
  Fout += 27 * (size_t)H * blockIdx.y; valid += 3 * (size_t)H * blockIdx.y; samples += 7 * (size_t)H * blockIdx.y;
  valid[3 * h] = valid[3 * h + 1] = valid[3 * h + 2] = 0;
  if (h >= (int)pb.seed) return;   // iterations getSubset found no sample for
  double A[7][9];
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const int id = samples[7 * (size_t)h + i];
    const double x0 = p1[2 * id], y0 = p1[2 * id + 1], x1 = p2[2 * id], y1 = p2[2 * id + 1];
    A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1; A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1; A[i][6] = x0; A[i][7] = y0; A[i][8] = 1.0;
  }
  // reduced row echelon form by Gauss-Jordan elimination with complete pivoting, everything unrolled and the run-time pivot position
  // applied by selects (as in k_f_hypotheses: the matrix stays in registers); the two columns never chosen are the free ones
  int colOf[7];
  unsigned usedMask = 0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    if (ok) {
      int pr = k, pc = -1;
      double best = 0;
#pragma unroll
      for (int i = k; i < 7; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const double a = fabs(A[i][j]);
          const bool take = !((usedMask >> j) & 1u) && a > best;
          best = take ? a : best; pr = take ? i : pr; pc = take ? j : pc;
        }
      if (pc < 0 || !(best > 0)) {
        ok = false;
      } else {
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const double tmp = A[k][j];
          double apr = tmp;
#pragma unroll
          for (int i = k + 1; i < 7; i++) apr = i == pr ? A[i][j] : apr;
          A[k][j] = apr;
#pragma unroll
          for (int i = k + 1; i < 7; i++) A[i][j] = i == pr ? tmp : A[i][j];
        }
        usedMask |= 1u << pc; colOf[k] = pc;
        double piv = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) piv = j == pc ? A[k][j] : piv;
        const double inv = 1.0 / piv;
#pragma unroll
        for (int j = 0; j < 9; j++) A[k][j] *= inv;
#pragma unroll
        for (int i = 0; i < 7; i++)
          if (i != k) {
            double fcoef = 0;
#pragma unroll
            for (int j = 0; j < 9; j++) fcoef = j == pc ? A[i][j] : fcoef;
            const bool nz = fcoef != 0.0;
#pragma unroll
            for (int j = 0; j < 9; j++) A[i][j] = nz ? A[i][j] - fcoef * A[k][j] : A[i][j];
          }
      }
    }
  }
  if (!ok) return;
  int fa = -1, fb = -1;
#pragma unroll
  for (int j = 0; j < 9; j++) if (!((usedMask >> j) & 1u)) { if (fa < 0) fa = j; else fb = j; }
  double f1[9], f2[9];
#pragma unroll
  for (int j = 0; j < 9; j++) { f1[j] = j == fa ? 1.0 : 0.0; f2[j] = j == fb ? 1.0 : 0.0; }
#pragma unroll
  for (int k = 0; k < 7; k++) {
    double aa = 0, ab = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) { aa = j == fa ? A[k][j] : aa; ab = j == fb ? A[k][j] : ab; }
#pragma unroll
    for (int j = 0; j < 9; j++) { f1[j] = j == colOf[k] ? -aa : f1[j]; f2[j] = j == colOf[k] ? -ab : f2[j]; }
  }
  // unit length (the elimination's basis has entries of very different size; OpenCV's comes from an SVD), then run7Point
  double n1 = 0, n2 = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) { n1 += f1[j] * f1[j]; n2 += f2[j] * f2[j]; }
  n1 = 1.0 / sqrt(n1); n2 = 1.0 / sqrt(n2);
#pragma unroll
  for (int j = 0; j < 9; j++) { f1[j] *= n1; f2[j] *= n2; }
#pragma unroll
  for (int j = 0; j < 9; j++) f1[j] -= f2[j];
  // det(lambda f1 + f2) = c0 lambda^3 + c1 lambda^2 + c2 lambda + c3
  double c[4];
  c[0] = det3rows(f1, f1 + 3, f1 + 6);
  c[1] = det3rows(f2, f1 + 3, f1 + 6) + det3rows(f1, f2 + 3, f1 + 6) + det3rows(f1, f1 + 3, f2 + 6);
  c[2] = det3rows(f1, f2 + 3, f2 + 6) + det3rows(f2, f1 + 3, f2 + 6) + det3rows(f2, f2 + 3, f1 + 6);
  c[3] = det3rows(f2, f2 + 3, f2 + 6);
  double r[3];
  const int nr = cv_solve_cubic(c, r);
  for (int k = 0; k < nr; k++) {
    double lambda = r[k], mu = 1.0;
    const double sden = f1[8] * r[k] + f2[8];
    double F[9];
    if (fabs(sden) > DBL_EPSILON) { mu = 1.0 / sden; lambda *= mu; F[8] = 1.0; } else F[8] = 0.0;
    bool fin = true;
#pragma unroll
    for (int j = 0; j < 8; j++) { F[j] = f1[j] * lambda + f2[j] * mu; fin = fin && isfinite(F[j]); }
    if (!fin) continue;
#pragma unroll
    for (int j = 0; j < 9; j++) Fout[9 * (size_t)(3 * h + k) + j] = F[j];
    valid[3 * h + k] = 1;
  }
}

// cv::LMeDSPointSetRegistrator::run over the models of k_f7_hypotheses — what cv::findFundamentalMat(FM_RANSAC) runs BELOW 15 points
// (calib3d ptsetreg.cpp, restated from the published algorithm): a fixed number of iterations (RANSACUpdateNumIters(confidence, 0.45, 7,
// maxIters), at least 3: 300 for 0.99), the same sample procedure (getSubset with 1000 attempts), every model of a sample in order;
// a model's score is the MEDIAN of its float errors (std::nth_element at count / 2: the upper median), a strictly smaller median
// replaces the best; then sigma = 2.5 * 1.4826 * (1 + 5 / (count - 7)) * sqrt(minMedian), at least 0.001, the inliers are the errors
// <= (float)(sigma^2), and the call succeeds when at least 7 of them remain (the mask is written either way).  One workgroup per
// problem (n <= 14 here): the threads take the models in turn, the lowest (median, index) wins, thread 0 writes the mask.
// sel[0] = best model (-1: none), sel[1] = iterations run, sel[2] = inliers, sel[3] = success.
__global__ __launch_bounds__(256) void k_lmeds_select(const float* __restrict__ p1, const float* __restrict__ p2, const RansacProb* __restrict__ probs, int H,
                                                      int niters, const double* __restrict__ Fall, const int* __restrict__ valid, int* __restrict__ sel,
                                                      unsigned char* __restrict__ mask, double* __restrict__ Fbest) {
  const RansacProb pb = probs[blockIdx.x];
  const int n = pb.n, tid = threadIdx.x;
  p1 += 2 * (size_t)pb.off; p2 += 2 * (size_t)pb.off; mask += pb.off;
  Fall += 27 * (size_t)H * blockIdx.x; valid += 3 * (size_t)H * blockIdx.x; sel += 4 * (size_t)blockIdx.x; Fbest += 9 * (size_t)blockIdx.x;
  const int iters = min(min(niters, H), (int)pb.seed);
  __shared__ unsigned long long s_best[256];
  unsigned long long best = ~0ull;   // (median's float bits << 32) | model index: non-negative floats order as their bit patterns
  for (int m = tid; m < 3 * iters; m += 256) {
    if (!valid[m]) continue;
    double F[9];
    for (int k = 0; k < 9; k++) F[k] = Fall[9 * (size_t)m + k];
    float e[16];
    for (int i = 0; i < 16; i++) e[i] = 3.0e38f;
    for (int i = 0; i < n && i < 16; i++) e[i] = (float)epi_err(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
    for (int i = 1; i < 16; i++) {   // insertion sort (n <= 14; NaN errors of a degenerate model sort as they fall: the model cannot win)
      const float v = e[i];
      int j = i - 1;
      while (j >= 0 && e[j] > v) { e[j + 1] = e[j]; j--; }
      e[j + 1] = v;
    }
    const float med = e[n / 2];
    if (!(med >= 0.0f)) continue;
    const unsigned long long key = ((unsigned long long)__float_as_uint(med) << 32) | (unsigned)m;
    best = key < best ? key : best;
  }
  s_best[tid] = best;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_best[tid] = s_best[tid + s] < s_best[tid] ? s_best[tid + s] : s_best[tid];
    __syncthreads();
  }
  if (tid != 0) return;
  best = s_best[0];
  sel[1] = iters;
  if (best == ~0ull) { sel[0] = -1; sel[2] = 0; sel[3] = 0; for (int i = 0; i < n; i++) mask[i] = 0; for (int k = 0; k < 9; k++) Fbest[k] = 0.0; return; }
  const int bm = (int)(unsigned)best;
  const double minMedian = (double)__uint_as_float((unsigned)(best >> 32));
  double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(minMedian);
  sigma = fmax(sigma, 0.001);
  const float t = (float)(sigma * sigma);
  double F[9];
  for (int k = 0; k < 9; k++) { F[k] = Fall[9 * (size_t)bm + k]; Fbest[k] = F[k]; }
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    const int in = (float)epi_err(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= t ? 1 : 0;
    mask[i] = (unsigned char)in; cnt += in;
  }
  sel[0] = bm; sel[2] = cnt; sel[3] = cnt >= 7 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// PnP: P3P hypotheses (Grunert 1841 as in Haralick et al., "Review and analysis of solutions of the three point perspective
// pose estimation problem", IJCV 1994): quartic in v = s3 / s1, then u = s2 / s1, the three depths, and the rigid motion that
// takes the object triangle onto the camera-frame triangle.
// ---------------------------------------------------------------------------------------------------------------------------
// real roots of a4 x^4 + a3 x^3 + a2 x^2 + a1 x + a0 (Ferrari through the resolvent cubic), each polished by Newton steps
__host__ __device__ __forceinline__ int quartic_real_roots(double a4, double a3, double a2, double a1, double a0, double* roots) {
  if (!(fabs(a4) > 1e-14 * (fabs(a3) + fabs(a2) + fabs(a1) + fabs(a0)))) return 0;
  const double b = a3 / a4, c = a2 / a4, d = a1 / a4, e = a0 / a4;
  // depressed quartic y^4 + p y^2 + q y + r, x = y - b / 4
  const double p = c - 3.0 * b * b / 8.0, q = d - b * c / 2.0 + b * b * b / 8.0, r = e - b * d / 4.0 + b * b * c / 16.0 - 3.0 * b * b * b * b / 256.0;
  int nr = 0;
  double ys[4];
  if (fabs(q) < 1e-14 * (1.0 + fabs(p) + fabs(r))) {  // biquadratic
    const double disc = p * p - 4.0 * r;
    if (disc >= -1e-9 * (p * p + fabs(4.0 * r))) {
      const double sd = sqrt(fmax(disc, 0.0));
      const double z[2] = {(-p + sd) / 2.0, (-p - sd) / 2.0};
      for (int k = 0; k < 2; k++) if (z[k] >= 0) { ys[nr++] = sqrt(z[k]); ys[nr++] = -sqrt(z[k]); }
    }
  } else {
    // resolvent cubic 8 m^3 + 8 p m^2 + (2 p^2 - 8 r) m - q^2 = 0: its largest real root is positive when q != 0
    const double B = p, C = (p * p - 4.0 * r) / 4.0, D = -q * q / 8.0;   // m^3 + B m^2 + C m + D
    const double Q3 = (3.0 * C - B * B) / 9.0, R3 = (9.0 * B * C - 27.0 * D - 2.0 * B * B * B) / 54.0;
    const double disc = Q3 * Q3 * Q3 + R3 * R3;
    double m;
    if (disc >= 0) {
      const double sd = sqrt(disc);
      m = cbrt(R3 + sd) + cbrt(R3 - sd) - B / 3.0;
    } else {
      const double th = acos(fmax(-1.0, fmin(1.0, R3 / sqrt(-Q3 * Q3 * Q3))));
      m = 2.0 * sqrt(-Q3) * cos(th / 3.0) - B / 3.0;
    }
    for (int k = 0; k < 3; k++) {  // polish the resolvent root
      const double fm = ((m + B) * m + C) * m + D, dfm = (3.0 * m + 2.0 * B) * m + C;
      if (dfm != 0.0) m -= fm / dfm;
    }
    if (m > 0) {
      const double s2m = sqrt(2.0 * m);
      // (y^2 + p/2 + m)^2 = 2 m (y - q / (4 m))^2  ->  y^2 + s y + (p/2 + m - q / (2 s)) = 0  and  y^2 - s y + (p/2 + m + q / (2 s)) = 0
      const double t2 = p / 2.0 + m - q / (2.0 * s2m), t1 = p / 2.0 + m + q / (2.0 * s2m);   // t2 pairs with the (-s) roots, t1 with (+s)
      // a double root shows up as a discriminant of either sign at rounding level: keep it (Newton below settles it), or a
      // fronto-parallel triangle — the planar scenes of an RGB-D camera facing a wall — loses its true pose
      const double dA = 2.0 * m - 4.0 * t2, dB = 2.0 * m - 4.0 * t1, tol = 1e-9 * (fabs(2.0 * m) + fabs(4.0 * t1) + fabs(4.0 * t2));
      if (dA >= -tol) { const double sd = sqrt(fmax(dA, 0.0)); ys[nr++] = (-s2m + sd) / 2.0; ys[nr++] = (-s2m - sd) / 2.0; }
      if (dB >= -tol) { const double sd = sqrt(fmax(dB, 0.0)); ys[nr++] = (s2m + sd) / 2.0; ys[nr++] = (s2m - sd) / 2.0; }
    }
  }
  for (int k = 0; k < nr; k++) {
    double x = ys[k] - b / 4.0;
    for (int itn = 0; itn < 3; itn++) {
      const double fx = (((a4 * x + a3) * x + a2) * x + a1) * x + a0, dfx = ((4.0 * a4 * x + 3.0 * a3) * x + 2.0 * a2) * x + a1;
      if (dfx == 0.0) break;
      x -= fx / dfx;
    }
    roots[k] = x;
  }
  return nr;
}

__host__ __device__ __forceinline__ void triad(const double* A0, const double* A1, const double* A2, double E[3][3], bool& ok) {
  double d1[3], d2[3], e3[3];
  for (int k = 0; k < 3; k++) { d1[k] = A1[k] - A0[k]; d2[k] = A2[k] - A0[k]; }
  const double n1 = sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
  e3[0] = d1[1] * d2[2] - d1[2] * d2[1]; e3[1] = d1[2] * d2[0] - d1[0] * d2[2]; e3[2] = d1[0] * d2[1] - d1[1] * d2[0];
  const double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
  ok = ok && n1 > 1e-12 && n3 > 1e-12;
  for (int k = 0; k < 3; k++) { E[k][0] = d1[k] / n1; E[k][2] = e3[k] / n3; }
  E[0][1] = E[1][2] * E[2][0] - E[2][2] * E[1][0];
  E[1][1] = E[2][2] * E[0][0] - E[0][2] * E[2][0];
  E[2][1] = E[0][2] * E[1][0] - E[1][2] * E[0][0];
}

// Grunert's P3P for one triangle: object points P (rows), unit bearings j (rows) -> up to 4 poses (R row-major 9, t 3 each) with
// x_cam = R X + t, in ascending order of the quartic's root
__host__ __device__ __forceinline__ int p3p_solve(const double P[3][3], const double j[3][3], double* poses /* 4 x 12 */) {
  auto dist2 = [&](int a, int b) { double s = 0; for (int k = 0; k < 3; k++) s += (P[a][k] - P[b][k]) * (P[a][k] - P[b][k]); return s; };
  auto dot = [&](int a, int b) { return j[a][0] * j[b][0] + j[a][1] * j[b][1] + j[a][2] * j[b][2]; };
  const double a2 = dist2(1, 2), b2 = dist2(0, 2), c2 = dist2(0, 1);
  const double ca = dot(1, 2), cb = dot(0, 2), cg = dot(0, 1);
  int nsol = 0;
  if (!(a2 > 1e-18 && b2 > 1e-18 && c2 > 1e-18)) return 0;
  const double q = (a2 - c2) / b2, w = (a2 + c2) / b2;
  const double A4 = (q - 1) * (q - 1) - 4 * c2 / b2 * ca * ca;
  const double A3 = 4 * (q * (1 - q) * cb - (1 - w) * ca * cg + 2 * c2 / b2 * ca * ca * cb);
  const double A2 = 2 * (q * q - 1 + 2 * q * q * cb * cb + 2 * (b2 - c2) / b2 * ca * ca - 4 * w * ca * cb * cg + 2 * (b2 - a2) / b2 * cg * cg);
  const double A1 = 4 * (-q * (1 + q) * cb + 2 * a2 / b2 * cg * cg * cb - (1 - w) * ca * cg);
  const double A0 = (1 + q) * (1 + q) - 4 * a2 / b2 * cg * cg;
  double roots[4];
  const int nr = quartic_real_roots(A4, A3, A2, A1, A0, roots);
  for (int a = 1; a < nr; a++) {   // canonical solution order (ascending v): ties between equally good poses resolve the same way
    const double key = roots[a];
    int b = a - 1;
    while (b >= 0 && roots[b] > key) { roots[b + 1] = roots[b]; b--; }
    roots[b + 1] = key;
  }
  for (int rI = 0; rI < nr && nsol < 4; rI++) {
    const double v = roots[rI];
    if (!(v > 0) || !isfinite(v)) continue;
    if (rI > 0 && fabs(v - roots[rI - 1]) <= 1e-9 * fabs(v)) continue;   // a double root gives one pose
    const double den = 2 * (cg - v * ca);
    if (fabs(den) < 1e-12) continue;
    const double u = ((-1 + q) * v * v - 2 * q * cb * v + 1 + q) / den;
    if (!(u > 0) || !isfinite(u)) continue;
    const double dd = 1 + u * u - 2 * u * cg;
    if (!(dd > 1e-18)) continue;
    const double s1 = sqrt(c2 / dd), s2 = u * s1, s3 = v * s1;
    double Q[3][3];
    for (int k = 0; k < 3; k++) { Q[0][k] = s1 * j[0][k]; Q[1][k] = s2 * j[1][k]; Q[2][k] = s3 * j[2][k]; }
    double EQ[3][3], EP[3][3];
    bool ok = true;
    triad(Q[0], Q[1], Q[2], EQ, ok);
    triad(P[0], P[1], P[2], EP, ok);
    if (!ok) continue;
    double* o = poses + 12 * nsol;
    double R[9];
    for (int a = 0; a < 3; a++) for (int bq = 0; bq < 3; bq++) R[3 * a + bq] = EQ[a][0] * EP[bq][0] + EQ[a][1] * EP[bq][1] + EQ[a][2] * EP[bq][2];
    bool fin = true;
    for (int k = 0; k < 9; k++) { o[k] = R[k]; fin = fin && isfinite(R[k]); }
    for (int a = 0; a < 3; a++) { o[9 + a] = Q[0][a] - (R[3 * a] * P[0][0] + R[3 * a + 1] * P[0][1] + R[3 * a + 2] * P[0][2]); fin = fin && isfinite(o[9 + a]); }
    if (!fin) continue;
    nsol++;
  }
  return nsol;
}

// one thread = one sample of 3 correspondences -> up to 4 poses; unused slots are marked invalid
__global__ __launch_bounds__(64) void k_p3p_hypotheses(const float* __restrict__ obj, const float* __restrict__ img, const RansacProb* __restrict__ probs, int H,
                                                       double fx, double fy, double cx, double cy, double* __restrict__ poses, int* __restrict__ valid) {
  const int h = blockIdx.x * 64 + threadIdx.x;
  if (h >= H) return;
  const RansacProb pb = probs[blockIdx.y];
  const int n = pb.n;
  obj += 3 * (size_t)pb.off; img += 2 * (size_t)pb.off;
  poses += 12 * (size_t)4 * H * blockIdx.y; valid += (size_t)4 * H * blockIdx.y;
  if (n < 4) { for (int s = 0; s < 4; s++) valid[4 * h + s] = 0; return; }
  int idx[3];
  sample_distinct<3>(pb.seed, h, n, idx);
  double P[3][3], j[3][3];
  for (int i = 0; i < 3; i++) {
    for (int k = 0; k < 3; k++) P[i][k] = obj[3 * idx[i] + k];
    const double bx = (img[2 * idx[i]] - cx) / fx, by = (img[2 * idx[i] + 1] - cy) / fy;
    const double nn = 1.0 / sqrt(bx * bx + by * by + 1.0);
    j[i][0] = bx * nn; j[i][1] = by * nn; j[i][2] = nn;
  }
  // the solutions go straight to their slots (a local array indexed by the running solution count lived in scratch memory)
  const int nsol = p3p_solve(P, j, poses + 12 * ((size_t)4 * h));
  for (int s = 0; s < 4; s++) valid[4 * h + s] = s < nsol ? 1 : 0;
}

__device__ __forceinline__ double reproj_err2(const double* R, const double* t, double fx, double fy, double cx, double cy, const float* X, const float* uv) {
  const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  if (!(z > 1e-9)) return 1e300;
  const double du = fx * x / z + cx - uv[0], dv = fy * y / z + cy - uv[1];
  return du * du + dv * dv;
}

__global__ __launch_bounds__(256) void k_pnp_score(const float* __restrict__ obj, const float* __restrict__ img, const RansacProb* __restrict__ probs, int H4,
                                                   const double* __restrict__ poses, const int* __restrict__ valid, double fx, double fy, double cx, double cy,
                                                   double thr2, int* __restrict__ counts) {
  const int h = blockIdx.x;
  const RansacProb pb = probs[blockIdx.y];
  const int n = pb.n;
  obj += 3 * (size_t)pb.off; img += 2 * (size_t)pb.off;
  poses += 12 * (size_t)H4 * blockIdx.y; valid += (size_t)H4 * blockIdx.y; counts += (size_t)H4 * blockIdx.y;
  if (!valid[h]) { if (threadIdx.x == 0) counts[h] = 0; return; }
  double Rt[12];
  for (int k = 0; k < 12; k++) Rt[k] = poses[12 * (size_t)h + k];
  int total = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    bool in = false;
    if (i < n) in = reproj_err2(Rt, Rt + 9, fx, fy, cx, cy, obj + 3 * i, img + 2 * i) <= thr2;
    total += block_count256(in);
  }
  if (threadIdx.x == 0) counts[h] = total;
}

__device__ __forceinline__ double block_sum256(double v, double* sm) {
  const int tid = threadIdx.x;
  __syncthreads();
  sm[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) sm[tid] += sm[tid + s]; __syncthreads(); }
  const double r = sm[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ void exp_so3(const double* w, double* E) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  const double A = th < 1e-8 ? 1.0 - th2 / 6.0 : sin(th) / th, B = th < 1e-8 ? 0.5 - th2 / 24.0 : (1.0 - cos(th)) / th2;
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double kk = 0;
      for (int c = 0; c < 3; c++) kk += K[3 * a + c] * K[3 * c + b];
      E[3 * a + b] = (a == b ? 1.0 : 0.0) + A * K[3 * a + b] + B * kk;
    }
}

// inlier mask of the selected pose, then Levenberg-Marquardt on the reprojection error over the inliers (what
// SOLVEPNP_ITERATIVE does), one workgroup.  out: rvec (Rodrigues) + tvec; inliers as ascending indices.
// Per problem (blockIdx.x): results record `res` = {nin, success, pad, pad (16 B) | rvec, tvec (48 B)}, inlier list at inliers + off.
__global__ __launch_bounds__(256) void k_pnp_refine(const float* __restrict__ obj, const float* __restrict__ img, const RansacProb* __restrict__ probs, int H4,
                                                    const double* __restrict__ poses, const int* __restrict__ sel, double fx, double fy, double cx, double cy,
                                                    double thr2, int* __restrict__ inliers, unsigned char* __restrict__ resAll) {
  const RansacProb pb = probs[blockIdx.x];
  const int n = pb.n;
  obj += 3 * (size_t)pb.off; img += 2 * (size_t)pb.off; inliers += pb.off;
  poses += 12 * (size_t)H4 * blockIdx.x; sel += 4 * (size_t)blockIdx.x;
  int* nin = reinterpret_cast<int*>(resAll + 64 * (size_t)blockIdx.x);
  int* success = nin + 1;
  double* rt = reinterpret_cast<double*>(resAll + 64 * (size_t)blockIdx.x + 16);
  __shared__ double sm[256];
  __shared__ double sm27[27][256];   // 54 KB: all normal-equation sums of an iteration in one reduction tree
  __shared__ double sR[9], st[3], sRn[9], stn[3], sH[36], sg[6], sd[6];
  __shared__ int s_cnt, s_wsum[5];
  __shared__ double s_lambda;
  __shared__ int s_stop;
  const int tid = threadIdx.x;
  const int best = sel[0];
  if (best < 0) { if (tid == 0) { *nin = 0; *success = 0; for (int k = 0; k < 6; k++) rt[k] = 0; } return; }
  if (tid < 9) sR[tid] = poses[12 * (size_t)best + tid];
  if (tid < 3) st[tid] = poses[12 * (size_t)best + 9 + tid];
  if (tid == 0) { s_cnt = 0; s_lambda = 1e-3; s_stop = 0; }
  __syncthreads();
  // ordered inlier list (ascending index): block scans over chunks of 256
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    const bool in = i < n && reproj_err2(sR, st, fx, fy, cx, cy, obj + 3 * i, img + 2 * i) <= thr2;
    int incl = in ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int tt = __shfl_up(incl, o); if ((tid & 63) >= o) incl += tt; }
    __syncthreads();
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    int base = s_cnt;
    for (int w = 0; w < (tid >> 6); w++) base += s_wsum[w];
    if (in) inliers[base + incl - 1] = i;
    __syncthreads();
    if (tid == 0) s_cnt += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    __syncthreads();
  }
  const int m = s_cnt;
  if (tid == 0) *nin = m;
  auto cost_of = [&](const double* R, const double* t) -> double {
    double c = 0;
    for (int e = tid; e < m; e += 256) { const int i = inliers[e]; c += fmin(reproj_err2(R, t, fx, fy, cx, cy, obj + 3 * i, img + 2 * i), 1e12); }
    return block_sum256(c, sm);
  };
  __threadfence_block();
  __syncthreads();
  double cost = cost_of(sR, st);
  for (int iter = 0; iter < 20 && m >= 3; iter++) {
    // normal equations of the current pose: left perturbation R <- exp(w) R, t <- t + dt
    double H[21], g[6];
    for (int k = 0; k < 21; k++) H[k] = 0;
    for (int k = 0; k < 6; k++) g[k] = 0;
    for (int e = tid; e < m; e += 256) {
      const int i = inliers[e];
      const float* X = obj + 3 * i;
      const double rx = sR[0] * X[0] + sR[1] * X[1] + sR[2] * X[2], ry = sR[3] * X[0] + sR[4] * X[1] + sR[5] * X[2], rz = sR[6] * X[0] + sR[7] * X[1] + sR[8] * X[2];
      const double x = rx + st[0], y = ry + st[1], z = rz + st[2];
      if (!(z > 1e-9)) continue;
      const double iz = 1.0 / z, du = fx * x * iz + cx - img[2 * i], dv = fy * y * iz + cy - img[2 * i + 1];
      // d(u,v)/d p_c
      const double a00 = fx * iz, a02 = -fx * x * iz * iz, a11 = fy * iz, a12 = -fy * y * iz * iz;
      // d p_c / d w = -[R X]_x ; d p_c / d t = I
      double Ju[6], Jv[6];
      // -[r]_x = [[0, rz, -ry], [-rz, 0, rx], [ry, -rx, 0]]
      Ju[0] = a02 * ry;            Ju[1] = a00 * rz - a02 * rx;   Ju[2] = -a00 * ry;
      Jv[0] = -a11 * rz + a12 * ry; Jv[1] = -a12 * rx;             Jv[2] = a11 * rx;
      Ju[3] = a00; Ju[4] = 0; Ju[5] = a02;
      Jv[3] = 0; Jv[4] = a11; Jv[5] = a12;
      int k = 0;
      for (int a = 0; a < 6; a++) {
        for (int b = a; b < 6; b++) H[k++] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
        g[a] += Ju[a] * du + Jv[a] * dv;
      }
    }
    {
      // the 27 sums through ONE tree (block_sum256's association for each, so the same bits; 27 passes of it were ~250 barriers per
      // iteration and most of this kernel's 165 us)
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 21; k++) sm27[k][tid] = H[k];
#pragma unroll
      for (int a = 0; a < 6; a++) sm27[21 + a][tid] = g[a];
      __syncthreads();
      for (int sft = 128; sft > 0; sft >>= 1) {
        if (tid < sft) {
#pragma unroll
          for (int k = 0; k < 27; k++) sm27[k][tid] += sm27[k][tid + sft];
        }
        __syncthreads();
      }
      if (tid == 0) {
        int k = 0;
        for (int a = 0; a < 6; a++)
          for (int b = a; b < 6; b++) { const double v = sm27[k++][0]; sH[6 * a + b] = v; sH[6 * b + a] = v; }
        for (int a = 0; a < 6; a++) sg[a] = sm27[21 + a][0];
      }
    }
    __syncthreads();
    bool accepted = false;
    for (int tries = 0; tries < 8 && !accepted; tries++) {
      if (tid == 0) {  // (H + lambda diag(H)) d = -g by Cholesky
        // every loop unrolled: with run-time subscripts the 6 x 6 system lived in scratch memory (one thread, ~100 dependent
        // scratch round trips per solve)
        double A[36], b6[6];
#pragma unroll
        for (int k = 0; k < 36; k++) A[k] = sH[k];
#pragma unroll
        for (int a = 0; a < 6; a++) { A[7 * a] += s_lambda * fmax(sH[7 * a], 1e-12); b6[a] = -sg[a]; }
        bool ok = true;
#pragma unroll
        for (int jx = 0; jx < 6; jx++) {
          if (ok) {
            double d = A[7 * jx];
#pragma unroll
            for (int k = 0; k < jx; k++) d -= A[6 * jx + k] * A[6 * jx + k];
            if (!(d > 0)) {
              ok = false;
            } else {
              d = sqrt(d);
              A[7 * jx] = d;
#pragma unroll
              for (int i = jx + 1; i < 6; i++) {
                double sacc = A[6 * i + jx];
#pragma unroll
                for (int k = 0; k < jx; k++) sacc -= A[6 * i + k] * A[6 * jx + k];
                A[6 * i + jx] = sacc / d;
              }
            }
          }
        }
        if (ok) {
#pragma unroll
          for (int i = 0; i < 6; i++) {
            double sacc = b6[i];
#pragma unroll
            for (int k = 0; k < i; k++) sacc -= A[6 * i + k] * b6[k];
            b6[i] = sacc / A[7 * i];
          }
#pragma unroll
          for (int i = 5; i >= 0; i--) {
            double sacc = b6[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) sacc -= A[6 * k + i] * b6[k];
            b6[i] = sacc / A[7 * i];
          }
          double E[9];
          exp_so3(b6, E);
          for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sRn[3 * a + b] = E[3 * a] * sR[b] + E[3 * a + 1] * sR[3 + b] + E[3 * a + 2] * sR[6 + b];
          for (int a = 0; a < 3; a++) stn[a] = st[a] + b6[3 + a];
          for (int a = 0; a < 6; a++) sd[a] = b6[a];
        } else {
          for (int k = 0; k < 9; k++) sRn[k] = sR[k];
          for (int k = 0; k < 3; k++) stn[k] = st[k];
          for (int a = 0; a < 6; a++) sd[a] = 0;
        }
      }
      __syncthreads();
      const double ncost = cost_of(sRn, stn);
      if (ncost < cost) {
        accepted = true;
        const double stepn = fabs(sd[0]) + fabs(sd[1]) + fabs(sd[2]) + fabs(sd[3]) + fabs(sd[4]) + fabs(sd[5]);
        __syncthreads();
        if (tid == 0) {
          for (int k = 0; k < 9; k++) sR[k] = sRn[k];
          for (int k = 0; k < 3; k++) st[k] = stn[k];
          s_lambda = fmax(s_lambda * 0.1, 1e-12);
          if (stepn < 1e-12 || cost - ncost <= 1e-14 * cost) s_stop = 1;
        }
        cost = ncost;
      } else {
        __syncthreads();
        if (tid == 0) s_lambda *= 10.0;
      }
      __syncthreads();
    }
    if (!accepted || s_stop) break;
  }
  __syncthreads();
  if (tid == 0) {
    // rotation matrix -> Rodrigues vector
    const double tr = sR[0] + sR[4] + sR[8];
    const double cth = fmax(-1.0, fmin(1.0, (tr - 1.0) / 2.0)), th = acos(cth);
    double w[3] = {sR[7] - sR[5], sR[2] - sR[6], sR[3] - sR[1]};
    const double sn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) / 2.0;   // sin(theta)
    if (sn > 1e-12) {
      const double f = th / (2.0 * sn);
      for (int k = 0; k < 3; k++) w[k] *= f;
    } else if (cth > 0) {
      for (int k = 0; k < 3; k++) w[k] *= 0.5;
    } else {  // theta = pi: axis from the diagonal
      const double ax[3] = {sqrt(fmax((sR[0] + 1) / 2, 0.0)), sqrt(fmax((sR[4] + 1) / 2, 0.0)), sqrt(fmax((sR[8] + 1) / 2, 0.0))};
      w[0] = th * ax[0]; w[1] = th * ax[1] * (sR[1] >= 0 ? 1 : -1); w[2] = th * ax[2] * (sR[2] >= 0 ? 1 : -1);
    }
    for (int k = 0; k < 3; k++) { rt[k] = w[k]; rt[3 + k] = st[k]; }
    *success = m > 0 ? 1 : 0;
  }
}

// one pass over `nprob` problems with the models of the first H iterations; unfinished[b] = 1 where the loop wanted more than H
// (lmeds: the problems are below 15 points — H = the fixed iteration count, k_lmeds_select instead of score / select / mask,
//  unfinished[b] = 1 where the call FAILED, i.e. fewer than 7 inliers: OpenCV returns an empty matrix then, the mask stays as written)
static dvs_status fm_cv_pass(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts1, const float* pts2, double threshold, double confidence,
                             int32_t max_iters, int32_t H, double* F9, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iterations, uint8_t* unfinished,
                             bool lmeds = false) {
  int maxn = 0;
  for (int b = 0; b < nprob; b++) maxn = std::max(maxn, offsets[b + 1] - offsets[b]);
  const int total = offsets[nprob];
  hipStream_t st = matcher_stream(ctx);
  const int H3 = 3 * H;
  const size_t hb = ((size_t)nprob * sizeof(RansacProb) + 15) & ~(size_t)15, pb = ((size_t)total * 8 + 15) & ~(size_t)15;
  const size_t sb = ((size_t)nprob * H * 7 * 4 + 15) & ~(size_t)15;
  const size_t inb = hb + 2 * pb + sb;
  const size_t fb = (size_t)nprob * H3 * 72, vb = (size_t)nprob * H3 * 4;
  const size_t outb = ((size_t)nprob * (16 + 72) + (size_t)total + 3) & ~(size_t)3;
  uint8_t* base;
  DVS_TRY(matcher_scratch(ctx, 0, inb + fb + 2 * vb + outb + 64, (void**)&base));
  const RansacProb* d_probs = (const RansacProb*)base;
  float* d_p1 = (float*)(base + hb); float* d_p2 = (float*)(base + hb + pb);
  const int32_t* d_samples = (const int32_t*)(base + hb + 2 * pb);
  double* d_F = (double*)(base + inb);
  int* d_valid = (int*)(base + inb + fb); int* d_counts = (int*)(base + inb + fb + vb);
  uint8_t* d_out = base + inb + fb + 2 * vb;
  int* d_sel = (int*)d_out; double* d_Fb = (double*)(d_out + (size_t)nprob * 16); unsigned char* d_mask = d_out + (size_t)nprob * 88;
  uint8_t* hio; int *hseq, *counter;
  DVS_TRY(matcher_pinned(ctx, inb + outb, (void**)&hio, &hseq, &counter));
  RansacProb* hp = (RansacProb*)hio;
  memcpy(hio + hb, pts1, (size_t)total * 8); memcpy(hio + hb + pb, pts2, (size_t)total * 8);
  int32_t* hs = (int32_t*)(hio + hb + 2 * pb);
  memset(hs, 0, sb);
  for (int b = 0; b < nprob; b++) {
    const int n = offsets[b + 1] - offsets[b];
    const int found = cv_subsets(pts1 + 2 * (size_t)offsets[b], pts2 + 2 * (size_t)offsets[b], n, 7, H, hs + (size_t)b * H * 7, lmeds ? 1000 : 10000);
    hp[b] = RansacProb{offsets[b], n, (unsigned long long)found};   // the seed field carries the iterations that have a sample
  }
  const int ndw_in = (int)(inb / 4);
  hipLaunchKernelGGL(k_io_import, dim3((ndw_in + 255) / 256), dim3(256), 0, st, (const uint32_t*)hio, (uint32_t*)base, ndw_in);
  hipLaunchKernelGGL(k_f7_hypotheses, dim3((H + 63) / 64, nprob), dim3(64), 0, st, d_p1, d_p2, d_probs, d_samples, H, d_F, d_valid);
  if (lmeds) {
    hipLaunchKernelGGL(k_lmeds_select, dim3(nprob), dim3(256), 0, st, d_p1, d_p2, d_probs, H, H, d_F, d_valid, d_sel, d_mask, d_Fb);
  } else {
    hipLaunchKernelGGL(k_f_score, dim3(H3, nprob), dim3(256), 0, st, d_p1, d_p2, d_probs, H3, d_F, d_valid, threshold * threshold, d_counts, 1);
    hipLaunchKernelGGL(k_ransac_select, dim3(nprob), dim3(1), 0, st, d_counts, H3, d_probs, 7, confidence, 3, d_sel, 1, max_iters);
    hipLaunchKernelGGL(k_f_mask, dim3((std::max(maxn, 9) + 255) / 256, nprob), dim3(256), 0, st, d_p1, d_p2, d_probs, H3, d_F, d_sel, threshold * threshold, d_mask, d_Fb, 1);
  }
  uint8_t* hout = hio + inb;
  if (outb <= 65536) {
    const int seq = ++*counter;
    hipLaunchKernelGGL(k_io_export, dim3(1), dim3(256), 0, st, (const uint32_t*)d_out, (uint32_t*)hout, (int)(outb / 4), hseq, seq);
    DVS_HIP(hipGetLastError());
    DVS_TRY(io_wait(hseq, seq, st));
  } else {
    DVS_HIP(hipGetLastError());
    DVS_HIP(hipMemcpyAsync(hout, d_out, outb, hipMemcpyDeviceToHost, st));
    DVS_HIP(hipStreamSynchronize(st));
  }
  const int* sel = (const int*)hout;
  for (int b = 0; b < nprob; b++) {
    if (n_inliers) n_inliers[b] = sel[4 * b] >= 0 ? sel[4 * b + 2] : 0;
    if (iterations) iterations[b] = sel[4 * b + 1];
    unfinished[b] = lmeds ? (uint8_t)(sel[4 * b + 3] ? 0 : 1) : (uint8_t)sel[4 * b + 3];
    if (F9) {
      if (lmeds && !sel[4 * b + 3]) memset(F9 + 9 * (size_t)b, 0, 72);
      else memcpy(F9 + 9 * (size_t)b, hout + (size_t)nprob * 16 + 72 * (size_t)b, 72);
    }
  }
  memcpy(inlier_mask, hout + (size_t)nprob * 88, (size_t)total);
  return DVS_OK;
}

// a subset of a batch's problems as a batch of its own (concatenated copies), and its results written back
struct FmSubset {
  std::vector<int> which;
  std::vector<int32_t> off;
  std::vector<float> q1, q2;
  std::vector<double> F;
  std::vector<uint8_t> mask, flag;
  std::vector<int32_t> nin, its;
  void gather(const std::vector<int>& w, const int32_t* offsets, const float* pts1, const float* pts2) {
    which = w;
    off.assign(w.size() + 1, 0);
    for (size_t i = 0; i < w.size(); i++) off[i + 1] = off[i] + (offsets[w[i] + 1] - offsets[w[i]]);
    q1.resize((size_t)off.back() * 2 + 2); q2.resize((size_t)off.back() * 2 + 2);
    for (size_t i = 0; i < w.size(); i++) {
      const int b = w[i], n = offsets[b + 1] - offsets[b];
      memcpy(q1.data() + 2 * (size_t)off[i], pts1 + 2 * (size_t)offsets[b], (size_t)n * 8);
      memcpy(q2.data() + 2 * (size_t)off[i], pts2 + 2 * (size_t)offsets[b], (size_t)n * 8);
    }
    F.assign(w.size() * 9, 0.0); mask.assign((size_t)off.back() + 1, 0); flag.assign(w.size(), 0); nin.assign(w.size(), 0); its.assign(w.size(), 0);
  }
  void scatter(const int32_t* offsets, double* F9, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iterations) const {
    for (size_t i = 0; i < which.size(); i++) {
      const int b = which[i], n = offsets[b + 1] - offsets[b];
      memcpy(inlier_mask + offsets[b], mask.data() + off[i], (size_t)n);
      if (n_inliers) n_inliers[b] = nin[i];
      if (iterations) iterations[b] = its[i];
      if (F9) memcpy(F9 + 9 * (size_t)b, F.data() + 9 * i, 72);
    }
  }
};


}  // namespace dvs

#include "pnp_cv.h"   // cv::solvePnPRansac by OpenCV's procedure: EPnP hypotheses, float scoring, solvePnP(ITERATIVE) refit (needs RansacProb, block_count256)

using namespace dvs;

// the 5-point samples of cv::solvePnPRansac's RANSAC stage: getSubset without a checkSubset (host: the sequence is sequential by nature)
static void cv_subsets_nocheck(int n, int modelPoints, int iters, int32_t* idx_out) {
  CvRng rng(~0ull);
  for (int it = 0; it < iters; it++)
    for (int i = 0; i < modelPoints; i++) {
      int v;
      bool dup;
      do {
        v = rng.uniform(0, n);
        dup = false;
        for (int j = 0; j < i; j++) dup = dup || idx_out[(size_t)it * modelPoints + j] == v;
      } while (dup);
      idx_out[(size_t)it * modelPoints + i] = v;
    }
}

extern "C" {

#ifdef DVS_TEST_HOOKS   // libdvslam_hip_test.so only (include/dvslam_hip_test.h)
// host-logic test hooks (no GPU): the product's quartic and P3P routines, compiled for the host
int32_t dvs_test_quartic_roots(double a4, double a3, double a2, double a1, double a0, double* roots4) {
  return quartic_real_roots(a4, a3, a2, a1, a0, roots4);
}
int32_t dvs_test_p3p(const double* P9, const double* j9, double* poses48) {
  double P[3][3], j[3][3];
  memcpy(P, P9, sizeof(P)); memcpy(j, j9, sizeof(j));
  return p3p_solve(P, j, poses48);
}
#endif  // DVS_TEST_HOOKS

// ---- batches of independent RANSAC problems (one launch sequence for all of them; the single-problem entry points are batches of one) ----
// problem b = correspondences [offsets[b], offsets[b + 1]) of the concatenated point arrays, sampler seed seeds[b]
dvs_status dvs_find_fundamental_ransac_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts1, const float* pts2, double threshold,
                                             double confidence, int32_t max_iters, const uint64_t* seeds, double* F9, uint8_t* inlier_mask,
                                             int32_t* n_inliers) {
  DVS_ARG(ctx && nprob >= 0 && max_iters >= 1 && max_iters <= 4096 && threshold > 0);
  if (nprob == 0) return DVS_OK;
  DVS_ARG(offsets && seeds && offsets[0] == 0);
  int maxn = 0;
  for (int b = 0; b < nprob; b++) { DVS_ARG(offsets[b + 1] >= offsets[b]); maxn = std::max(maxn, offsets[b + 1] - offsets[b]); }
  const int total = offsets[nprob];
  DVS_ARG(total == 0 || (pts1 && pts2 && inlier_mask));
  if (n_inliers) memset(n_inliers, 0, (size_t)nprob * 4);
  if (F9) memset(F9, 0, (size_t)nprob * 72);
  if (total) memset(inlier_mask, 0, (size_t)total);
  if (maxn < 8) return DVS_OK;   // cv::findFundamentalMat needs >= 7 points for FM_RANSAC (8 for our kernel): empty F, masks of zeros
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  const int H = max_iters;
  const size_t hb = ((size_t)nprob * sizeof(RansacProb) + 15) & ~(size_t)15, pb = ((size_t)total * 8 + 15) & ~(size_t)15;
  const size_t inb = hb + 2 * pb;
  const size_t fb = (size_t)nprob * H * 72, vb = (size_t)nprob * H * 4;
  const size_t outb = ((size_t)nprob * (16 + 72) + (size_t)total + 3) & ~(size_t)3;
  uint8_t* base;
  DVS_TRY(matcher_scratch(ctx, 0, inb + fb + 2 * vb + outb + 64, (void**)&base));
  const RansacProb* d_probs = (const RansacProb*)base;
  float* d_p1 = (float*)(base + hb); float* d_p2 = (float*)(base + hb + pb);
  double* d_F = (double*)(base + inb);
  int* d_valid = (int*)(base + inb + fb); int* d_counts = (int*)(base + inb + fb + vb);
  uint8_t* d_out = base + inb + fb + 2 * vb;                       // [sel 16 x nprob | Fb 72 x nprob | mask total]: the pinned block's layout
  int* d_sel = (int*)d_out; double* d_Fb = (double*)(d_out + (size_t)nprob * 16); unsigned char* d_mask = d_out + (size_t)nprob * 88;
  uint8_t* hio; int *hseq, *counter;
  DVS_TRY(matcher_pinned(ctx, inb + outb, (void**)&hio, &hseq, &counter));
  RansacProb* hp = (RansacProb*)hio;
  for (int b = 0; b < nprob; b++) hp[b] = RansacProb{offsets[b], offsets[b + 1] - offsets[b], (unsigned long long)seeds[b]};
  memcpy(hio + hb, pts1, (size_t)total * 8); memcpy(hio + hb + pb, pts2, (size_t)total * 8);
  const int ndw_in = (int)(inb / 4);
  hipLaunchKernelGGL(k_io_import, dim3((ndw_in + 255) / 256), dim3(256), 0, st, (const uint32_t*)hio, (uint32_t*)base, ndw_in);
  hipLaunchKernelGGL(k_f_hypotheses, dim3((H + 63) / 64, nprob), dim3(64), 0, st, d_p1, d_p2, d_probs, H, d_F, d_valid);
  hipLaunchKernelGGL(k_f_score, dim3(H, nprob), dim3(256), 0, st, d_p1, d_p2, d_probs, H, d_F, d_valid, threshold * threshold, d_counts, 0);
  hipLaunchKernelGGL(k_ransac_select, dim3(nprob), dim3(1), 0, st, d_counts, H, d_probs, 8, confidence, 1, d_sel);
  hipLaunchKernelGGL(k_f_mask, dim3((std::max(maxn, 9) + 255) / 256, nprob), dim3(256), 0, st, d_p1, d_p2, d_probs, H, d_F, d_sel, threshold * threshold, d_mask, d_Fb, 0);
  uint8_t* hout = hio + inb;
  if (outb <= 65536) {   // small results leave through the export kernel + a polled sequence number (no copy command, no wake-up)
    const int seq = ++*counter;
    hipLaunchKernelGGL(k_io_export, dim3(1), dim3(256), 0, st, (const uint32_t*)d_out, (uint32_t*)hout, (int)(outb / 4), hseq, seq);
    DVS_HIP(hipGetLastError());
    DVS_TRY(io_wait(hseq, seq, st));
  } else {
    DVS_HIP(hipGetLastError());
    DVS_HIP(hipMemcpyAsync(hout, d_out, outb, hipMemcpyDeviceToHost, st));
    DVS_HIP(hipStreamSynchronize(st));
  }
  const int* sel = (const int*)hout;
  for (int b = 0; b < nprob; b++) {
    if (n_inliers) n_inliers[b] = sel[4 * b] >= 0 ? sel[4 * b + 2] : 0;
    if (F9) memcpy(F9 + 9 * (size_t)b, hout + (size_t)nprob * 16 + 72 * (size_t)b, 72);
  }
  if (total) memcpy(inlier_mask, hout + (size_t)nprob * 88, (size_t)total);
  return DVS_OK;
}

// cv::findFundamentalMat(FM_RANSAC) the way OpenCV 4.x runs it (see k_f7_hypotheses): RANSAC from 15 correspondences on, LMedS for 8..14
// (OpenCV switches there: k_lmeds_select); fewer than 8 are DVS_ERR_UNSUPPORTED (the reference calls with >= 8, frontend.cpp:627)
dvs_status dvs_find_fundamental_cv_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts1, const float* pts2, double threshold,
                                         double confidence, int32_t max_iters, double* F9, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iterations) {
  DVS_ARG(ctx && nprob >= 0 && max_iters >= 1 && max_iters <= 4096);
  if (nprob == 0) return DVS_OK;
  DVS_ARG(offsets && offsets[0] == 0 && pts1 && pts2 && inlier_mask);
  if (threshold <= 0) threshold = 3;                                                 // as cv::findFundamentalMat
  if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
  std::vector<int> big, small;
  for (int b = 0; b < nprob; b++) {
    DVS_ARG(offsets[b + 1] >= offsets[b]);
    const int n = offsets[b + 1] - offsets[b];
    if (n < 8) { set_error("dvs_find_fundamental_cv: problem %d has %d correspondences (the reference calls with >= 8, frontend.cpp:627)", b, n); return DVS_ERR_UNSUPPORTED; }
    (n >= 15 ? big : small).push_back(b);
  }
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  // ---- >= 15 points: RANSAC.  The adaptive rule ends most loops after a few dozen iterations (10 % outliers: ~10), and the sample
  // sequence is the same whatever the number of samples drawn: first the models of 96 iterations for every problem, then — only for
  // the problems whose loop wanted more — all max_iters (the same result a full run gives, by construction: a prefix of the same sequence).
  if (!big.empty()) {
    FmSubset A;
    A.gather(big, offsets, pts1, pts2);
    const int H1 = std::min<int>(max_iters, 96);
    DVS_TRY(fm_cv_pass(ctx, (int)big.size(), A.off.data(), A.q1.data(), A.q2.data(), threshold, confidence, max_iters, H1, A.F.data(), A.mask.data(), A.nin.data(),
                       A.its.data(), A.flag.data()));
    A.scatter(offsets, F9, inlier_mask, n_inliers, iterations);
    std::vector<int> again;
    for (size_t i = 0; i < big.size(); i++) if (A.flag[i]) again.push_back(big[i]);
    if (!again.empty() && H1 < max_iters) {
      FmSubset B;
      B.gather(again, offsets, pts1, pts2);
      DVS_TRY(fm_cv_pass(ctx, (int)again.size(), B.off.data(), B.q1.data(), B.q2.data(), threshold, confidence, max_iters, max_iters, B.F.data(), B.mask.data(),
                         B.nin.data(), B.its.data(), B.flag.data()));
      B.scatter(offsets, F9, inlier_mask, n_inliers, iterations);
    }
  }
  // ---- 8 .. 14 points: LMedS with its fixed iteration count (k_lmeds_select)
  if (!small.empty()) {
    const int niters = std::max(ransac_update_iters_host(confidence, 0.45, 7, max_iters), 3);   // outlierRatio = 0.45; 300 for 0.99
    FmSubset C;
    C.gather(small, offsets, pts1, pts2);
    DVS_TRY(fm_cv_pass(ctx, (int)small.size(), C.off.data(), C.q1.data(), C.q2.data(), threshold, confidence, max_iters, niters, C.F.data(), C.mask.data(), C.nin.data(),
                       C.its.data(), C.flag.data(), true));
    C.scatter(offsets, F9, inlier_mask, n_inliers, iterations);
  }
  return DVS_OK;
}

dvs_status dvs_find_fundamental_cv(dvs_matcher* ctx, const float* pts1, const float* pts2, int32_t n, double threshold, double confidence,
                                   int32_t max_iters, double* F9, uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iterations) {
  DVS_ARG(ctx && n >= 0);
  const int32_t offsets[2] = {0, n};
  return dvs_find_fundamental_cv_batch(ctx, 1, offsets, pts1, pts2, threshold, confidence, max_iters, F9, inlier_mask, n_inliers, iterations);
}

// host only (no GPU): the sample sequence of the call above — iteration it draws idx[7 it .. 7 it + 6]; *found = iterations that have one
dvs_status dvs_cv_ransac_subsets(const float* pts1, const float* pts2, int32_t n, int32_t model_points, int32_t iterations, int32_t* idx, int32_t* found) {
  DVS_ARG(idx && found && n >= model_points && model_points >= 1 && model_points <= 16 && iterations >= 0 && (pts1 == nullptr) == (pts2 == nullptr));
  if (!pts1) { cv_subsets_nocheck(n, model_points, iterations, idx); *found = iterations; }   // a callback without checkSubset (solvePnPRansac)
  else *found = cv_subsets(pts1, pts2, n, model_points, iterations, idx);
  return DVS_OK;
}

// cv::solvePnPRansac(obj, img, K, no distortion, rvec, tvec, false, iterations, reproj_err, confidence, inliers) as OpenCV 4.x runs it with
// its default flags (csrc/pnp_cv.h).  Problems with fewer than 6 points are refused per problem (success 0; the reference returns before
// the call, frontend.cpp:900).  iterations_run[b] = iterations the adaptive loop used (may be NULL).
dvs_status dvs_solve_pnp_ransac_cv_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts3d, const float* pts2d, const double* K4,
                                         int32_t iterations, double reproj_err, double confidence, double* rvec3, double* tvec3, int32_t* inliers,
                                         int32_t* n_inliers, int32_t* success, int32_t* iterations_run) {
  DVS_ARG(ctx && nprob >= 0 && iterations >= 1 && iterations <= 1024 && K4 && reproj_err > 0 && confidence > 0 && confidence < 1);
  if (nprob == 0) return DVS_OK;
  DVS_ARG(offsets && rvec3 && tvec3 && success && offsets[0] == 0);
  for (int b = 0; b < nprob; b++) DVS_ARG(offsets[b + 1] >= offsets[b]);
  const int total = offsets[nprob];
  DVS_ARG(total == 0 || (pts3d && pts2d));
  memset(success, 0, (size_t)nprob * 4);
  if (n_inliers) memset(n_inliers, 0, (size_t)nprob * 4);
  if (iterations_run) memset(iterations_run, 0, (size_t)nprob * 4);
  memset(rvec3, 0, (size_t)nprob * 24); memset(tvec3, 0, (size_t)nprob * 24);
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  const int H = iterations;
  const size_t hb = ((size_t)nprob * sizeof(RansacProb) + 15) & ~(size_t)15, ob = ((size_t)total * 12 + 15) & ~(size_t)15, ib = ((size_t)total * 8 + 15) & ~(size_t)15;
  const size_t sb = ((size_t)nprob * H * 5 * 4 + 15) & ~(size_t)15;
  const size_t inb = hb + ob + ib + sb;
  const size_t mb = (size_t)nprob * H * 18 * 8, vb = (size_t)nprob * H * 4, selb = (size_t)nprob * 16;
  const size_t outb = (size_t)nprob * 64 + (size_t)total * 4 + selb;      // [result records 64 x nprob | inlier lists 4 x total | select records]
  uint8_t* base;
  DVS_TRY(matcher_scratch(ctx, 0, inb + mb + vb + outb + 64, (void**)&base));
  const RansacProb* d_probs = (const RansacProb*)base;
  float* d_obj = (float*)(base + hb); float* d_img = (float*)(base + hb + ob);
  const int* d_samples = (const int*)(base + hb + ob + ib);
  double* d_models = (double*)(base + inb);
  int* d_counts = (int*)(base + inb + mb);
  uint8_t* d_out = base + inb + mb + vb;
  int* d_inl = (int*)(d_out + (size_t)nprob * 64);
  int* d_sel = (int*)(d_out + (size_t)nprob * 64 + (size_t)total * 4);
  uint8_t* hio; int *hseq, *counter;
  DVS_TRY(matcher_pinned(ctx, inb + outb, (void**)&hio, &hseq, &counter));
  RansacProb* hp = (RansacProb*)hio;
  int32_t* hs = (int32_t*)(hio + hb + ob + ib);
  memset(hs, 0, sb);
  for (int b = 0; b < nprob; b++) {
    const int n = offsets[b + 1] - offsets[b];
    const bool run = n >= 6;     // (n == 5 would be solvePnP on all points, n == 4 the P3P kernel: not what the reference can reach)
    if (run) cv_subsets_nocheck(n, 5, H, hs + (size_t)b * H * 5);
    hp[b] = RansacProb{offsets[b], n, (unsigned long long)(run ? H : 0)};   // the seed field: iterations that have a sample
  }
  memcpy(hio + hb, pts3d, (size_t)total * 12); memcpy(hio + hb + ob, pts2d, (size_t)total * 8);
  const int ndw_in = (int)(inb / 4);
  hipLaunchKernelGGL(k_io_import, dim3((ndw_in + 255) / 256), dim3(256), 0, st, (const uint32_t*)hio, (uint32_t*)base, ndw_in);
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  const float thr = (float)(reproj_err * reproj_err);
  hipLaunchKernelGGL(k_epnp_hypotheses, dim3((H + kEpnpGroups - 1) / kEpnpGroups, nprob), dim3(64), 0, st, d_obj, d_img, d_probs, d_samples, H, fx, fy, cx, cy, d_models);
  hipLaunchKernelGGL(k_pnpcv_score, dim3(H, nprob), dim3(256), 0, st, d_obj, d_img, d_probs, H, d_models, fx, fy, cx, cy, thr, d_counts);
  hipLaunchKernelGGL(k_ransac_select, dim3(nprob), dim3(1), 0, st, d_counts, H, d_probs, 5, confidence, 1, d_sel, 1, H);
  hipLaunchKernelGGL(k_pnpcv_refit, dim3(nprob), dim3(64), 0, st, d_obj, d_img, d_probs, H, d_models, d_sel, fx, fy, cx, cy, thr, d_inl, d_out);
  uint8_t* hout = hio + inb;
  if (outb <= 65536) {
    const int seq = ++*counter;
    hipLaunchKernelGGL(k_io_export, dim3(1), dim3(256), 0, st, (const uint32_t*)d_out, (uint32_t*)hout, (int)(outb / 4), hseq, seq);
    DVS_HIP(hipGetLastError());
    DVS_TRY(io_wait(hseq, seq, st));
  } else {
    DVS_HIP(hipGetLastError());
    DVS_HIP(hipMemcpyAsync(hout, d_out, outb, hipMemcpyDeviceToHost, st));
    DVS_HIP(hipStreamSynchronize(st));
  }
  const int* hsel = (const int*)(hout + (size_t)nprob * 64 + (size_t)total * 4);
  for (int b = 0; b < nprob; b++) {
    int nin = 0, succ = 0;
    memcpy(&nin, hout + 64 * (size_t)b, 4); memcpy(&succ, hout + 64 * (size_t)b + 4, 4);
    if (n_inliers) n_inliers[b] = nin;
    if (iterations_run) iterations_run[b] = hsel[4 * b + 1];
    success[b] = succ;
    if (hsel[4 * b] >= 0) { memcpy(rvec3 + 3 * (size_t)b, hout + 64 * (size_t)b + 16, 24); memcpy(tvec3 + 3 * (size_t)b, hout + 64 * (size_t)b + 40, 24); }
    if (inliers && nin > 0) memcpy(inliers + offsets[b], hout + (size_t)nprob * 64 + 4 * (size_t)offsets[b], (size_t)nin * 4);
  }
  return DVS_OK;
}

dvs_status dvs_solve_pnp_ransac_cv(dvs_matcher* ctx, const float* pts3d, const float* pts2d, int32_t n, const double* K4, int32_t iterations,
                                   double reproj_err, double confidence, double* rvec3, double* tvec3, int32_t* inliers, int32_t* n_inliers,
                                   int32_t* success, int32_t* iterations_run) {
  DVS_ARG(ctx && n >= 0 && rvec3 && tvec3 && success);
  const int32_t offsets[2] = {0, n};
  return dvs_solve_pnp_ransac_cv_batch(ctx, 1, offsets, pts3d, pts2d, K4, iterations, reproj_err, confidence, rvec3, tvec3, inliers, n_inliers, success,
                                       iterations_run);
}

dvs_status dvs_find_fundamental_ransac(dvs_matcher* ctx, const float* pts1, const float* pts2, int32_t n, double threshold, double confidence,
                                       int32_t max_iters, uint64_t seed, double* F9, uint8_t* inlier_mask, int32_t* n_inliers) {
  DVS_ARG(ctx && n >= 0 && (inlier_mask || n == 0));
  const int32_t offsets[2] = {0, n};
  return dvs_find_fundamental_ransac_batch(ctx, 1, offsets, pts1, pts2, threshold, confidence, max_iters, &seed, F9, inlier_mask, n_inliers);
}

// inliers: concatenated like the points (problem b's ascending inlier indices at inliers + offsets[b], n_inliers[b] of them)
dvs_status dvs_solve_pnp_ransac_batch(dvs_matcher* ctx, int32_t nprob, const int32_t* offsets, const float* pts3d, const float* pts2d, const double* K4,
                                      int32_t iterations, double reproj_err, double confidence, const uint64_t* seeds, double* rvec3, double* tvec3,
                                      int32_t* inliers, int32_t* n_inliers, int32_t* success) {
  DVS_ARG(ctx && nprob >= 0 && iterations >= 1 && iterations <= 1024 && K4 && reproj_err > 0);
  if (nprob == 0) return DVS_OK;
  DVS_ARG(offsets && seeds && rvec3 && tvec3 && success && offsets[0] == 0);
  int maxn = 0;
  for (int b = 0; b < nprob; b++) { DVS_ARG(offsets[b + 1] >= offsets[b]); maxn = std::max(maxn, offsets[b + 1] - offsets[b]); }
  const int total = offsets[nprob];
  DVS_ARG(total == 0 || (pts3d && pts2d));
  memset(success, 0, (size_t)nprob * 4);
  if (n_inliers) memset(n_inliers, 0, (size_t)nprob * 4);
  memset(rvec3, 0, (size_t)nprob * 24); memset(tvec3, 0, (size_t)nprob * 24);
  if (maxn < 4) return DVS_OK;   // cv::solvePnPRansac: "npoints >= 4"
  DVS_HIP(hipSetDevice(matcher_device(ctx)));
  hipStream_t st = matcher_stream(ctx);
  const int H = iterations, H4 = 4 * H;
  const size_t hb = ((size_t)nprob * sizeof(RansacProb) + 15) & ~(size_t)15, ob = ((size_t)total * 12 + 15) & ~(size_t)15, ib = ((size_t)total * 8 + 15) & ~(size_t)15;
  const size_t inb = hb + ob + ib;
  const size_t posb = (size_t)nprob * H4 * 96, vb = (size_t)nprob * H4 * 4, selb = (size_t)nprob * 16;
  const size_t outb = (size_t)nprob * 64 + (size_t)total * 4;      // [result records 64 x nprob | inlier lists 4 x total]
  uint8_t* base;
  DVS_TRY(matcher_scratch(ctx, 0, inb + posb + 2 * vb + selb + outb + 64, (void**)&base));
  const RansacProb* d_probs = (const RansacProb*)base;
  float* d_obj = (float*)(base + hb); float* d_img = (float*)(base + hb + ob);
  double* d_poses = (double*)(base + inb);
  int* d_valid = (int*)(base + inb + posb); int* d_counts = (int*)(base + inb + posb + vb);
  int* d_sel = (int*)(base + inb + posb + 2 * vb);
  uint8_t* d_out = base + inb + posb + 2 * vb + selb;
  int* d_inl = (int*)(d_out + (size_t)nprob * 64);
  uint8_t* hio; int *hseq, *counter;
  DVS_TRY(matcher_pinned(ctx, inb + outb, (void**)&hio, &hseq, &counter));
  RansacProb* hp = (RansacProb*)hio;
  for (int b = 0; b < nprob; b++) hp[b] = RansacProb{offsets[b], offsets[b + 1] - offsets[b], (unsigned long long)seeds[b]};
  memcpy(hio + hb, pts3d, (size_t)total * 12); memcpy(hio + hb + ob, pts2d, (size_t)total * 8);
  const int ndw_in = (int)(inb / 4);
  hipLaunchKernelGGL(k_io_import, dim3((ndw_in + 255) / 256), dim3(256), 0, st, (const uint32_t*)hio, (uint32_t*)base, ndw_in);
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3], thr2 = reproj_err * reproj_err;
  hipLaunchKernelGGL(k_p3p_hypotheses, dim3((H + 63) / 64, nprob), dim3(64), 0, st, d_obj, d_img, d_probs, H, fx, fy, cx, cy, d_poses, d_valid);
  hipLaunchKernelGGL(k_pnp_score, dim3(H4, nprob), dim3(256), 0, st, d_obj, d_img, d_probs, H4, d_poses, d_valid, fx, fy, cx, cy, thr2, d_counts);
  hipLaunchKernelGGL(k_ransac_select, dim3(nprob), dim3(1), 0, st, d_counts, H4, d_probs, 3, confidence, 4, d_sel);
  hipLaunchKernelGGL(k_pnp_refine, dim3(nprob), dim3(256), 0, st, d_obj, d_img, d_probs, H4, d_poses, d_sel, fx, fy, cx, cy, thr2, d_inl, d_out);
  uint8_t* hout = hio + inb;
  if (outb <= 65536) {
    const int seq = ++*counter;
    hipLaunchKernelGGL(k_io_export, dim3(1), dim3(256), 0, st, (const uint32_t*)d_out, (uint32_t*)hout, (int)(outb / 4), hseq, seq);
    DVS_HIP(hipGetLastError());
    DVS_TRY(io_wait(hseq, seq, st));
  } else {
    DVS_HIP(hipGetLastError());
    DVS_HIP(hipMemcpyAsync(hout, d_out, outb, hipMemcpyDeviceToHost, st));
    DVS_HIP(hipStreamSynchronize(st));
  }
  for (int b = 0; b < nprob; b++) {
    int nin = 0, succ = 0;
    memcpy(&nin, hout + 64 * (size_t)b, 4); memcpy(&succ, hout + 64 * (size_t)b + 4, 4);
    if (n_inliers) n_inliers[b] = nin;
    success[b] = succ;
    if (succ) { memcpy(rvec3 + 3 * (size_t)b, hout + 64 * (size_t)b + 16, 24); memcpy(tvec3 + 3 * (size_t)b, hout + 64 * (size_t)b + 40, 24); }
    if (inliers && nin > 0) memcpy(inliers + offsets[b], hout + (size_t)nprob * 64 + 4 * (size_t)offsets[b], (size_t)nin * 4);
  }
  return DVS_OK;
}

dvs_status dvs_solve_pnp_ransac(dvs_matcher* ctx, const float* pts3d, const float* pts2d, int32_t n, const double* K4, int32_t iterations,
                                double reproj_err, double confidence, uint64_t seed, double* rvec3, double* tvec3, int32_t* inliers,
                                int32_t* n_inliers, int32_t* success) {
  DVS_ARG(ctx && n >= 0 && rvec3 && tvec3 && success);
  const int32_t offsets[2] = {0, n};
  return dvs_solve_pnp_ransac_batch(ctx, 1, offsets, pts3d, pts2d, K4, iterations, reproj_err, confidence, &seed, rvec3, tvec3, inliers, n_inliers, success);
}

}  // extern "C"
