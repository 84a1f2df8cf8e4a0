// pnp_cv.h — device code of dvs_solve_pnp_ransac_cv: cv::solvePnPRansac as OpenCV 4.x runs it for the reference's call
// (frontend.cpp:911-921: flags = SOLVEPNP_ITERATIVE, 100 iterations, 4 px, 0.99).  Included by ransac.hip only.
//
// OpenCV's procedure (calib3d solvepnp.cpp / ptsetreg.cpp / epnp.cpp / calibration.cpp, restated from the published sources — the
// library is not in this image, PARITY UNPINNED):
//   RANSACPointSetRegistrator(modelPoints = 5) with ONE cv::RNG((uint64)-1): 5 distinct indices per iteration (made on the HOST, the
//   sequence is sequential by nature; no checkSubset for this callback) -> solvePnP(SOLVEPNP_EPNP) on the 5 points (k_epnp_hypotheses,
//   one thread per iteration) -> projectPoints into float, float squared error <= (float)(thr * thr) (k_pnpcv_score, a workgroup per
//   iteration) -> the loop replayed over the counts (k_ransac_select, modelPoints = 5) -> solvePnP(SOLVEPNP_ITERATIVE) on the inliers of
//   the best model (k_pnpcv_refit, one wavefront per problem: planar test, homography or DLT initialisation, CvLevMarq).
// Every SVD is a cyclic Jacobi eigen-decomposition of the (symmetric) normal matrix; cv::findHomography's Levenberg-Marquardt polish of
// the planar initialisation is not restated (the 20 CvLevMarq iterations that follow absorb it).
#pragma once

namespace dvs {
namespace pnpcv {

// ---- small dense routines (per thread, arrays in private memory) ------------------------------------------------------------------
// cyclic Jacobi on the symmetric N x N matrix A (row-major, destroyed); eigenvectors in the COLUMNS of V
template <int N>
__device__ inline void jacobi_eig(double* A, double* V) {
  for (int i = 0; i < N * N; i++) V[i] = 0.0;
  for (int i = 0; i < N; i++) V[i * N + i] = 1.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0, dg = 0.0;
    for (int p = 0; p < N; p++) { dg += A[p * N + p] * A[p * N + p]; for (int q = p + 1; q < N; q++) off += A[p * N + q] * A[p * N + q]; }
    if (!(off > 1e-28 * dg)) break;   // sums of squares: off-diagonal norm below 1e-14 of the diagonal's; Jacobi converges quadratically
    for (int p = 0; p < N - 1; p++)
      for (int q = p + 1; q < N; q++) {
        const double apq = A[p * N + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; k++) { const double a = A[k * N + p], b = A[k * N + q]; A[k * N + p] = c * a - s * b; A[k * N + q] = s * a + c * b; }
        for (int k = 0; k < N; k++) { const double a = A[p * N + k], b = A[q * N + k]; A[p * N + k] = c * a - s * b; A[q * N + k] = s * a + c * b; }
        for (int k = 0; k < N; k++) { const double a = V[k * N + p], b = V[k * N + q]; V[k * N + p] = c * a - s * b; V[k * N + q] = s * a + c * b; }
      }
  }
}
// The same decomposition by a GROUP of 16 lanes of one wavefront on matrices in LDS (N <= 16): lane k < N carries index k of the three
// inner loops of a rotation (columns p, q of A; rows p, q of A; columns p, q of V) — the 66 rotations of a sweep stay sequential, each
// costs three LDS round trips instead of 3 N.  Element for element the arithmetic of jacobi_eig: the same eigenvectors bit for bit.
// gl = lane & 15; every lane of the group must call it (lanes of other groups of the wavefront may diverge).
__device__ __forceinline__ void group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double group16_sum(double v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 16);
  return v;
}
template <int N>
__device__ inline void jacobi_eig_group(double* A, double* V, int gl) {
  for (int i = gl; i < N * N; i += 16) V[i] = (i / N == i % N) ? 1.0 : 0.0;
  group_sync();
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0, dg = 0.0;
    if (gl < N) { dg = A[gl * N + gl] * A[gl * N + gl]; for (int q = gl + 1; q < N; q++) off += A[gl * N + q] * A[gl * N + q]; }
    off = group16_sum(off); dg = group16_sum(dg);
    if (!(off > 1e-28 * dg)) break;
    for (int p = 0; p < N - 1; p++)
      for (int q = p + 1; q < N; q++) {
        const double apq = A[p * N + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        group_sync();     // every lane has read the three entries the rotation is made of
        if (gl < N) { const double a = A[gl * N + p], b = A[gl * N + q]; A[gl * N + p] = c * a - s * b; A[gl * N + q] = s * a + c * b; }
        group_sync();
        if (gl < N) {
          const double a = A[p * N + gl], b = A[q * N + gl]; A[p * N + gl] = c * a - s * b; A[q * N + gl] = s * a + c * b;
          const double va = V[gl * N + p], vb = V[gl * N + q]; V[gl * N + p] = c * va - s * vb; V[gl * N + q] = s * va + c * vb;
        }
        group_sync();
      }
  }
  group_sync();
}

// order[i] = index of the i-th LARGEST diagonal entry (stable)
template <int N>
__device__ inline void order_desc(const double* A, int* order) {
  for (int i = 0; i < N; i++) order[i] = i;
  for (int i = 1; i < N; i++) {   // insertion sort: stable
    const int oi = order[i];
    const double v = A[oi * N + oi];
    int j = i - 1;
    while (j >= 0 && A[order[j] * N + order[j]] < v) { order[j + 1] = order[j]; j--; }
    order[j + 1] = oi;
  }
}
// 6 x 6 symmetric positive definite solve, fully unrolled (registers); false when a pivot is not safely positive
__device__ inline bool chol6_solve(const double* A, const double* b, double* x) {
  double L[6][6], y[6];
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) dmax = fmax(dmax, A[7 * i]);
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double d = A[7 * j];
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < j) d -= L[j][k] * L[j][k];
    if (!(d > 1e-13 * dmax)) return false;
    d = sqrt(d);
    L[j][j] = d;
#pragma unroll
    for (int i = 0; i < 6; i++)
      if (i > j) {
        double s = A[6 * i + j];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) s -= L[i][k] * L[j][k];
        L[i][j] = s / d;
      }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < i) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int k = 0; k < 6; k++) if (k > i) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  return true;
}

// minimum-norm least squares of the M x N system (N <= 6) through the eigen-decomposition of A^T A: x = sum_k v_k (v_k . A^T b) / lambda_k over
// singular values sqrt(lambda_k) above 2 eps sum(w) (cvSolve(CV_SVD) / SVD::backSubst's threshold)
template <int M, int N>
__device__ inline void lstsq_svd(const double* A, const double* b, double* x) {
  double G[N * N], V[N * N], atb[N];
  for (int i = 0; i < N; i++) {
    double s = 0.0;
    for (int k = 0; k < M; k++) s += A[k * N + i] * b[k];
    atb[i] = s;
    for (int j = 0; j < N; j++) { double g = 0.0; for (int k = 0; k < M; k++) g += A[k * N + i] * A[k * N + j]; G[i * N + j] = g; }
  }
  jacobi_eig<N>(G, V);
  double sw = 0.0;
  for (int i = 0; i < N; i++) sw += sqrt(fmax(G[i * N + i], 0.0));
  const double thr = 2 * 2.220446049250313e-16 * sw;
  for (int j = 0; j < N; j++) x[j] = 0.0;
  for (int k = 0; k < N; k++) {
    const double lam = G[k * N + k];
    if (!(sqrt(fmax(lam, 0.0)) > thr)) continue;
    double s = 0.0;
    for (int i = 0; i < N; i++) s += V[i * N + k] * atb[i];
    s /= lam;
    for (int j = 0; j < N; j++) x[j] += V[j * N + k] * s;
  }
}
__device__ inline double det3(const double* R) {
  return R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
}
// nearest rotation factor U V^T of a 3 x 3 matrix (its SVD's): A V diag(1 / w) V^T with the third left vector from the cross product when
// the matrix is singular to rounding
__device__ inline void polar_uvt(const double* A, double* R) {
  double G[9], V[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) G[3 * i + j] = A[i] * A[j] + A[3 + i] * A[3 + j] + A[6 + i] * A[6 + j];
  jacobi_eig<3>(G, V);
  int o[3];
  order_desc<3>(G, o);
  double w[3], U[9];   // U columns
  for (int k = 0; k < 3; k++) {
    w[k] = sqrt(fmax(G[o[k] * 3 + o[k]], 0.0));
    for (int r = 0; r < 3; r++) {
      const double s = A[3 * r] * V[o[k]] + A[3 * r + 1] * V[3 + o[k]] + A[3 * r + 2] * V[6 + o[k]];
      U[3 * r + k] = w[k] > 0 ? s / w[k] : 0.0;
    }
  }
  if (!(w[2] > 1e-14 * w[0])) { U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * V[3 * j + o[0]] + U[3 * i + 1] * V[3 * j + o[1]] + U[3 * i + 2] * V[3 * j + o[2]];
}
// cv::Rodrigues vector -> matrix (+ the 3 x 9 Jacobian dR(k) / dr(i) at J[9 i + k])
__device__ inline void rodrigues_vec(const double* rv, double* R, double* J) {
  const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 2.220446049250313e-16) {
    for (int k = 0; k < 9; k++) R[k] = I[k];
    if (J) { for (int k = 0; k < 27; k++) J[k] = 0.0; J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
    return;
  }
  const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
  const double r[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
  const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
  const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
  if (J) {
    const double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0, 0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0, 0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
    const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
      const double ri = r[i];
      const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri, a4 = s * itheta;
      for (int k = 0; k < 9; k++) J[9 * i + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[9 * i + k] + a3 * rx[k] + a4 * drx[9 * i + k];
    }
  }
}
// cv::Rodrigues matrix -> vector (the matrix replaced by its nearest rotation first)
__device__ inline void rodrigues_mat(const double* Rin, double* rv) {
  double R[9];
  polar_uvt(Rin, R);
  double r[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { rv[0] = rv[1] = rv[2] = 0; return; }
    double t;
    t = (R[0] + 1) * 0.5; r[0] = sqrt(fmax(t, 0.));
    t = (R[4] + 1) * 0.5; r[1] = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
    t = (R[8] + 1) * 0.5; r[2] = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
    if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
    theta /= sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; k++) rv[k] = r[k] * theta;
    return;
  }
  const double vth = 1 / (2 * s) * theta;
  for (int k = 0; k < 3; k++) rv[k] = r[k] * vth;
}

// ---- class epnp on exactly 5 correspondences -----------------------------------------------------------------------------------------
struct Epnp5 {
  double fu, fv, uc, vc;
  double pws[15], us[10], alphas[20], pcs[15], cws[4][3], ccs[4][3];
};
__device__ inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline double dist2(const double* p, const double* q) { return (p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]); }

// epnp::qr_solve on the 6 x 4 system (its pivot scan never looks at the last row — kept)
__device__ inline void epnp_qr_solve(double* pA, double* pb, double* pX) {
  const int nr = 6, nc = 4;
  double A1[4], A2[4];
  for (int k = 0; k < nc; k++) {
    double eta = fabs(pA[k * nc + k]);
    for (int i = k + 1; i < nr; i++) { const double elt = fabs(pA[(i - 1) * nc + k]); if (eta < elt) eta = elt; }
    if (eta == 0) { for (int j = 0; j < nc; j++) pX[j] = 0.0; return; }
    double sum2 = 0.0;
    const double inv_eta = 1. / eta;
    for (int i = k; i < nr; i++) { pA[i * nc + k] *= inv_eta; sum2 += pA[i * nc + k] * pA[i * nc + k]; }
    double sigma = sqrt(sum2);
    if (pA[k * nc + k] < 0) sigma = -sigma;
    pA[k * nc + k] += sigma;
    A1[k] = sigma * pA[k * nc + k];
    A2[k] = -eta * sigma;
    for (int j = k + 1; j < nc; j++) {
      double sum = 0;
      for (int i = k; i < nr; i++) sum += pA[i * nc + k] * pA[i * nc + j];
      const double tau = sum / A1[k];
      for (int i = k; i < nr; i++) pA[i * nc + j] -= tau * pA[i * nc + k];
    }
  }
  for (int j = 0; j < nc; j++) {
    double tau = 0;
    for (int i = j; i < nr; i++) tau += pA[i * nc + j] * pb[i];
    tau /= A1[j];
    for (int i = j; i < nr; i++) pb[i] -= tau * pA[i * nc + j];
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
  for (int i = nc - 2; i >= 0; i--) {
    double sum = 0;
    for (int j = i + 1; j < nc; j++) sum += pA[i * nc + j] * pX[j];
    pX[i] = (pb[i] - sum) / A2[i];
  }
}

__device__ inline void epnp_gauss_newton(const double* L, const double* rho, double* betas) {
  for (int it = 0; it < 5; it++) {
    double A[24], b[6], x[4];
    for (int i = 0; i < 6; i++) {
      const double* rl = L + 10 * i;
      A[4 * i] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
      A[4 * i + 1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
      A[4 * i + 2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
      A[4 * i + 3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
      b[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] + rl[3] * betas[0] * betas[2] +
                       rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] + rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] +
                       rl[8] * betas[2] * betas[3] + rl[9] * betas[3] * betas[3]);
    }
    epnp_qr_solve(A, b, x);
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
}

// compute_R_and_t: camera-frame control points from the betas, point coordinates, sign, absolute orientation, mean reprojection error
__device__ inline double epnp_R_and_t(Epnp5& e, const double* ut, const double* betas, double* R, double* t) {
  for (int i = 0; i < 4; i++) e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0.0;
  for (int i = 0; i < 4; i++) {
    const double* v = ut + 12 * (11 - i);
    for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) e.ccs[j][k] += betas[i] * v[3 * j + k];
  }
  for (int i = 0; i < 5; i++) {
    const double* a = e.alphas + 4 * i;
    for (int j = 0; j < 3; j++) e.pcs[3 * i + j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
  }
  if (e.pcs[2] < 0.0) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) e.ccs[i][j] = -e.ccs[i][j];
    for (int i = 0; i < 15; i++) e.pcs[i] = -e.pcs[i];
  }
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) { pc0[j] += e.pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
  for (int j = 0; j < 3; j++) { pc0[j] /= 5; pw0[j] /= 5; }
  double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 3; j++) {
      abt[3 * j] += (e.pcs[3 * i + j] - pc0[j]) * (e.pws[3 * i] - pw0[0]);
      abt[3 * j + 1] += (e.pcs[3 * i + j] - pc0[j]) * (e.pws[3 * i + 1] - pw0[1]);
      abt[3 * j + 2] += (e.pcs[3 * i + j] - pc0[j]) * (e.pws[3 * i + 2] - pw0[2]);
    }
  polar_uvt(abt, R);
  if (det3(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
  t[0] = pc0[0] - dot3(R, pw0); t[1] = pc0[1] - dot3(R + 3, pw0); t[2] = pc0[2] - dot3(R + 6, pw0);
  double sum2 = 0.0;
  for (int i = 0; i < 5; i++) {
    const double* pw = e.pws + 3 * i;
    const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
    const double ue = e.uc + e.fu * Xc * inv_Zc, ve = e.vc + e.fv * Yc * inv_Zc;
    const double u = e.us[2 * i], v = e.us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / 5;
}

// solvePnP(SOLVEPNP_EPNP) on 5 float correspondences -> rvec, tvec
// wsA / wsV: two 144-double work arrays (LDS: the 12 x 12 eigen-decomposition is a long chain of dependent loads and stores — in
// private (scratch) memory the whole call took 19 ms per 100 hypotheses)
// gl: the lane's index in its group of 16 — lane 0 runs the (serial) algorithm, all 16 the 12 x 12 eigen-decomposition
__device__ inline void epnp5(const float* obj, const float* img, const int* idx, double fx, double fy, double cx, double cy, double* rvec, double* tvec,
                             double* wsA, double* wsV, int gl) {
  Epnp5 e;
  if (gl == 0) {
  e.fu = fx; e.fv = fy; e.uc = cx; e.vc = cy;
  const double ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < 5; i++) {
    const int s = idx[i];
    for (int k = 0; k < 3; k++) e.pws[3 * i + k] = obj[3 * s + k];
    const float xn = (float)(((double)img[2 * s] - cx) * ifx), yn = (float)(((double)img[2 * s + 1] - cy) * ify);   // undistortPoints into CV_32FC2
    e.us[2 * i] = xn * fx + cx;
    e.us[2 * i + 1] = yn * fy + cy;
  }
  // control points: centroid + the principal axes of the object points
  for (int j = 0; j < 3; j++) { double s = 0; for (int i = 0; i < 5; i++) s += e.pws[3 * i + j]; e.cws[0][j] = s / 5; }
  {
    double m[9], V[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { double s = 0; for (int i = 0; i < 5; i++) s += (e.pws[3 * i + a] - e.cws[0][a]) * (e.pws[3 * i + b] - e.cws[0][b]); m[3 * a + b] = s; }
    jacobi_eig<3>(m, V);
    int o[3];
    order_desc<3>(m, o);
    for (int i = 1; i < 4; i++) {
      const double k = sqrt(fmax(m[o[i - 1] * 3 + o[i - 1]], 0.0) / 5);
      for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * V[3 * j + o[i - 1]];
    }
  }
  // barycentric coordinates: pseudo-inverse of the control-point offsets (cvInvert(CV_SVD))
  {
    double cc[9], G[9], V[9], ci[9];
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) G[3 * i + j] = cc[i] * cc[j] + cc[3 + i] * cc[3 + j] + cc[6 + i] * cc[6 + j];
    jacobi_eig<3>(G, V);
    double sw = 0;
    for (int k = 0; k < 3; k++) sw += sqrt(fmax(G[4 * k], 0.0));
    const double thr = 2 * 2.220446049250313e-16 * sw;
    for (int k = 0; k < 9; k++) ci[k] = 0.0;
    for (int k = 0; k < 3; k++) {            // pinv = sum_k v_k v_k^T / lambda_k * cc^T
      const double lam = G[4 * k];
      if (!(sqrt(fmax(lam, 0.0)) > thr)) continue;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          const double vtct = V[k] * cc[3 * b] + V[3 + k] * cc[3 * b + 1] + V[6 + k] * cc[3 * b + 2];   // (v_k^T cc^T)[b]
          ci[3 * a + b] += V[3 * a + k] * vtct / lam;
        }
    }
    for (int i = 0; i < 5; i++) {
      const double* pi = e.pws + 3 * i;
      double* a = e.alphas + 4 * i;
      for (int j = 0; j < 3; j++) a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) + ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }
  // M^T M and its eigenvectors, smallest last (rows of ut)
  {
    double* mtm = wsA;
    for (int k = 0; k < 144; k++) mtm[k] = 0.0;
    for (int i = 0; i < 5; i++) {
      double M1[12], M2[12];
      const double* as = e.alphas + 4 * i;
      for (int k = 0; k < 4; k++) {
        M1[3 * k] = as[k] * fx; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (cx - e.us[2 * i]);
        M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * fy; M2[3 * k + 2] = as[k] * (cy - e.us[2 * i + 1]);
      }
      for (int a = 0; a < 12; a++) for (int b = 0; b < 12; b++) mtm[12 * a + b] += M1[a] * M1[b] + M2[a] * M2[b];
    }
  }
  }   // gl == 0
  jacobi_eig_group<12>(wsA, wsV, gl);
  if (gl != 0) return;
  double* ut = wsA;
  {
    int o[12];
    order_desc<12>(wsA, o);
    for (int i = 0; i < 12; i++) for (int k = 0; k < 12; k++) ut[12 * i + k] = wsV[12 * k + o[i]];   // (the eigenvalues are used up: ut overwrites mtm)
  }
  double L[60], rho[6];
  {
    const double* v[4] = {ut + 132, ut + 120, ut + 108, ut + 96};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = L + 10 * i;
      row[0] = dot3(dv[0][i], dv[0][i]); row[1] = 2.0f * dot3(dv[0][i], dv[1][i]); row[2] = dot3(dv[1][i], dv[1][i]);
      row[3] = 2.0f * dot3(dv[0][i], dv[2][i]); row[4] = 2.0f * dot3(dv[1][i], dv[2][i]); row[5] = dot3(dv[2][i], dv[2][i]);
      row[6] = 2.0f * dot3(dv[0][i], dv[3][i]); row[7] = 2.0f * dot3(dv[1][i], dv[3][i]); row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
      row[9] = dot3(dv[3][i], dv[3][i]);
    }
    rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]); rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]); rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);
  }
  double bestErr = 0.0, Rb[9], tb[3];
  for (int N = 1; N <= 3; N++) {
    double betas[4] = {0, 0, 0, 0};
    if (N == 1) {                                   // [B11 B12 B13 B14]
      double l[24], b4[4];
      for (int i = 0; i < 6; i++) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
      lstsq_svd<6, 4>(l, rho, b4);
      if (b4[0] < 0) { betas[0] = sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
      else { betas[0] = sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0]; }
    } else if (N == 2) {                            // [B11 B12 B22]
      double l[18], b3[3];
      for (int i = 0; i < 6; i++) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
      lstsq_svd<6, 3>(l, rho, b3);
      if (b3[0] < 0) { betas[0] = sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
      else { betas[0] = sqrt(b3[0]); betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
      if (b3[1] < 0) betas[0] = -betas[0];
    } else {                                        // [B11 B12 B22 B13 B23]
      double l[30], b5[5];
      for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) l[5 * i + k] = L[10 * i + k];
      lstsq_svd<6, 5>(l, rho, b5);
      if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
      else { betas[0] = sqrt(b5[0]); betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
      if (b5[1] < 0) betas[0] = -betas[0];
      betas[2] = b5[3] / betas[0];
    }
    epnp_gauss_newton(L, rho, betas);
    double R[9], t[3];
    const double err = epnp_R_and_t(e, ut, betas, R, t);
    if (N == 1 || err < bestErr) { bestErr = err; for (int k = 0; k < 9; k++) Rb[k] = R[k]; for (int k = 0; k < 3; k++) tb[k] = t[k]; }
  }
  rodrigues_mat(Rb, rvec);
  for (int k = 0; k < 3; k++) tvec[k] = tb[k];
}

// PnPRansacCallback::computeError of one point: projectPoints into float, float squared distance
__device__ inline float proj_err_f(const double* R, const double* t, double fx, double fy, double cx, double cy, const float* Xf, const float* uv) {
  const double X = Xf[0], Y = Xf[1], Z = Xf[2];
  double x = R[0] * X + R[1] * Y + R[2] * Z + t[0], y = R[3] * X + R[4] * Y + R[5] * Z + t[1], z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
  z = z ? 1. / z : 1;
  x *= z; y *= z;
  const float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
  const float dx = uv[0] - px, dy = uv[1] - py;
  float s = 0;
  s += dx * dx;
  s += dy * dy;
  return s;
}

}  // namespace pnpcv

// one thread per RANSAC iteration: EPnP on its 5-point sample -> model (rvec, tvec) and the model's rotation matrix
constexpr int kEpnpGroups = 4;   // hypotheses per workgroup (one wavefront): 16 lanes and 2 x 145 doubles of LDS each
__global__ __launch_bounds__(64) void k_epnp_hypotheses(const float* __restrict__ obj, const float* __restrict__ img, const RansacProb* __restrict__ probs,
                                                        const int* __restrict__ samples, int H, double fx, double fy, double cx, double cy,
                                                        double* __restrict__ models /* [H][18]: rvec, tvec, R, pad */) {
  __shared__ double wsA[kEpnpGroups][145], wsV[kEpnpGroups][145];
  const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
  const int h = blockIdx.x * kEpnpGroups + grp;
  if (h >= H) return;
  const RansacProb pb = probs[blockIdx.y];
  obj += 3 * (size_t)pb.off; img += 2 * (size_t)pb.off;
  samples += 5 * ((size_t)H * blockIdx.y + h);
  double* m = models + 18 * ((size_t)H * blockIdx.y + h);
  if (h >= (int)pb.seed) { if (gl == 0) for (int k = 0; k < 18; k++) m[k] = 0.0; return; }   // (the seed field: iterations that have a sample)
  int idx[5];
  for (int k = 0; k < 5; k++) idx[k] = samples[k];
  double rv[3], tv[3], R[9];
  pnpcv::epnp5(obj, img, idx, fx, fy, cx, cy, rv, tv, wsA[grp], wsV[grp], gl);
  if (gl != 0) return;
  pnpcv::rodrigues_vec(rv, R, nullptr);     // projectPoints starts from the ROTATION VECTOR of the model
  for (int k = 0; k < 3; k++) { m[k] = rv[k]; m[3 + k] = tv[k]; }
  for (int k = 0; k < 9; k++) m[6 + k] = R[k];
}

// a workgroup per iteration: its model's inlier count by OpenCV's float comparison
__global__ __launch_bounds__(256) void k_pnpcv_score(const float* __restrict__ obj, const float* __restrict__ img, const RansacProb* __restrict__ probs, int H,
                                                     const double* __restrict__ models, double fx, double fy, double cx, double cy, float thr,
                                                     int* __restrict__ counts) {
  const int h = blockIdx.x;
  const RansacProb pb = probs[blockIdx.y];
  const int n = pb.n;
  obj += 3 * (size_t)pb.off; img += 2 * (size_t)pb.off;
  const double* m = models + 18 * ((size_t)H * blockIdx.y + h);
  counts += (size_t)H * blockIdx.y;
  if (h >= (int)pb.seed) { if (threadIdx.x == 0) counts[h] = 0; return; }
  double R[9], t[3];
  for (int k = 0; k < 9; k++) R[k] = m[6 + k];
  for (int k = 0; k < 3; k++) t[k] = m[3 + k];
  int total = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    bool in = false;
    if (i < n) in = pnpcv::proj_err_f(R, t, fx, fy, cx, cy, obj + 3 * i, img + 2 * i) <= thr;
    total += block_count256(in);
  }
  if (threadIdx.x == 0) counts[h] = total;
}

// sum over the 64 lanes, the same value in every lane (butterfly: fixed order)
__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// One wavefront per problem: the inliers of the best model in index order, then solvePnP(SOLVEPNP_ITERATIVE) on them —
// cvFindExtrinsicCameraParams2: planar test, homography / DLT initialisation, CvLevMarq(6, 2 count, (20, FLT_EPSILON)) on the analytic
// projection Jacobian.  Sums over the inliers are lane-strided partial sums folded by a butterfly (every lane then holds the same
// total and runs the same small dense arithmetic: no broadcasts).  out record per problem (64 B): {int nInliers, int success, pad,
// double rvec[3], tvec[3]}.
__global__ __launch_bounds__(64) void k_pnpcv_refit(const float* __restrict__ obj, const float* __restrict__ img, const RansacProb* __restrict__ probs, int H,
                                                    const double* __restrict__ models, const int* __restrict__ sel, double fx, double fy, double cx,
                                                    double cy, float thr, int* __restrict__ inl, unsigned char* __restrict__ out) {
  using namespace pnpcv;
  __shared__ double wsA[145], wsV[145], wsOut[16];   // the 9 x 9 / 12 x 12 eigen-decompositions of the initialisation: lane 0, in LDS
  const RansacProb pb = probs[blockIdx.x];
  const int n = pb.n, lane = threadIdx.x;
  obj += 3 * (size_t)pb.off; img += 2 * (size_t)pb.off; inl += pb.off;
  out += 64 * (size_t)blockIdx.x;
  const int best = sel[4 * blockIdx.x];
  if (best < 0) { if (lane < 16) ((int*)out)[lane] = 0; return; }
  const double* bm = models + 18 * ((size_t)H * blockIdx.x + best);
  double R[9], t[3];
  for (int k = 0; k < 9; k++) R[k] = bm[6 + k];
  for (int k = 0; k < 3; k++) t[k] = bm[3 + k];
  // inlier list in index order (ballot + prefix)
  int count = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < n && proj_err_f(R, t, fx, fy, cx, cy, obj + 3 * i, img + 2 * i) <= thr;
    const unsigned long long b = __ballot(in);
    if (in) inl[count + __popcll(b & ((1ull << lane) - 1ull))] = i;
    count += __popcll(b);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double r[3] = {bm[0], bm[1], bm[2]}, tt[3] = {t[0], t[1], t[2]};   // result <= 0: the RANSAC stage's model
  int success = 0;
  const double ifx = 1. / fx, ify = 1. / fy;
  // ---- initialisation ----
  double Mc[3] = {0, 0, 0};
  for (int e = lane; e < count; e += 64) for (int k = 0; k < 3; k++) Mc[k] += (double)obj[3 * inl[e] + k];
  for (int k = 0; k < 3; k++) Mc[k] = wave_allsum(Mc[k]) / count;
  double MM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = lane; e < count; e += 64) {
    const float* X = obj + 3 * inl[e];
    const double d[3] = {X[0] - Mc[0], X[1] - Mc[1], X[2] - Mc[2]};
    for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) MM[3 * a + b] += d[a] * d[b];
  }
  for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) { MM[3 * a + b] = wave_allsum(MM[3 * a + b]); MM[3 * b + a] = MM[3 * a + b]; }
  double Vm[9];
  int o3[3];
  {
    double G[9];
    for (int k = 0; k < 9; k++) G[k] = MM[k];
    jacobi_eig<3>(G, Vm);
    order_desc<3>(G, o3);
    for (int k = 0; k < 9; k++) MM[k] = G[k];
  }
  const double W1 = MM[4 * o3[1]], W2 = MM[4 * o3[2]];
  bool init_ok = count >= 4;
  if (init_ok && W2 / W1 < 1e-3) {
    // planar structure: homography between the plane's coordinates and the normalised image points
    double Rt[9];   // rows = right singular vectors
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = Vm[3 * b + o3[a]];
    if (Rt[2] * Rt[2] + Rt[5] * Rt[5] < 1e-10) { for (int k = 0; k < 9; k++) Rt[k] = (k % 4 == 0) ? 1.0 : 0.0; }
    if (det3(Rt) < 0) for (int k = 0; k < 9; k++) Rt[k] = -Rt[k];
    double Tt[3];
    for (int a = 0; a < 3; a++) Tt[a] = -(Rt[3 * a] * Mc[0] + Rt[3 * a + 1] * Mc[1] + Rt[3 * a + 2] * Mc[2]);
    auto plane_xy = [&](int e, double& X, double& Y, double& x, double& y) {   // cv::findHomography works on CV_32F copies of both point sets
      const float* P = obj + 3 * inl[e];
      X = (float)(Rt[0] * P[0] + Rt[1] * P[1] + Rt[2] * P[2] + Tt[0]); Y = (float)(Rt[3] * P[0] + Rt[4] * P[1] + Rt[5] * P[2] + Tt[1]);
      x = (float)(((double)img[2 * inl[e]] - cx) * ifx); y = (float)(((double)img[2 * inl[e] + 1] - cy) * ify);
    };
    double c4[4] = {0, 0, 0, 0}, s4[4] = {0, 0, 0, 0};   // cm.x, cm.y, cM.x, cM.y
    for (int e = lane; e < count; e += 64) { double X, Y, x, y; plane_xy(e, X, Y, x, y); c4[0] += x; c4[1] += y; c4[2] += X; c4[3] += Y; }
    for (int k = 0; k < 4; k++) c4[k] = wave_allsum(c4[k]) / count;
    for (int e = lane; e < count; e += 64) { double X, Y, x, y; plane_xy(e, X, Y, x, y); s4[0] += fabs(x - c4[0]); s4[1] += fabs(y - c4[1]); s4[2] += fabs(X - c4[2]); s4[3] += fabs(Y - c4[3]); }
    for (int k = 0; k < 4; k++) s4[k] = wave_allsum(s4[k]);
    init_ok = !(fabs(s4[0]) < 2.220446049250313e-16 || fabs(s4[1]) < 2.220446049250313e-16 || fabs(s4[2]) < 2.220446049250313e-16 || fabs(s4[3]) < 2.220446049250313e-16);
    if (init_ok) {
      for (int k = 0; k < 4; k++) s4[k] = count / s4[k];
      double LtL[81];
      for (int k = 0; k < 81; k++) LtL[k] = 0.0;
      for (int e = lane; e < count; e += 64) {
        double X, Y, x, y;
        plane_xy(e, X, Y, x, y);
        x = (x - c4[0]) * s4[0]; y = (y - c4[1]) * s4[1]; X = (X - c4[2]) * s4[2]; Y = (Y - c4[3]) * s4[3];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) LtL[9 * j + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
      }
      for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) { LtL[9 * j + k] = wave_allsum(LtL[9 * j + k]); LtL[9 * k + j] = LtL[9 * j + k]; }
      if (lane == 0) for (int k = 0; k < 81; k++) wsA[k] = LtL[k];
      group_sync();
      if (lane < 16) jacobi_eig_group<9>(wsA, wsV, lane);
      if (lane == 0) {
        int o9[9];
        order_desc<9>(wsA, o9);
        for (int k = 0; k < 9; k++) wsOut[k] = wsV[9 * k + o9[8]];
      }
      group_sync();
      double H0[9], Ht[9], h[9];
      for (int k = 0; k < 9; k++) H0[k] = wsOut[k];
      const double invHnorm[9] = {1. / s4[0], 0, c4[0], 0, 1. / s4[1], c4[1], 0, 0, 1};
      const double Hnorm2[9] = {s4[2], 0, -c4[2] * s4[2], 0, s4[3], -c4[3] * s4[3], 0, 0, 1};
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Ht[3 * a + b] = invHnorm[3 * a] * H0[b] + invHnorm[3 * a + 1] * H0[3 + b] + invHnorm[3 * a + 2] * H0[6 + b];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) h[3 * a + b] = Ht[3 * a] * Hnorm2[b] + Ht[3 * a + 1] * Hnorm2[3 + b] + Ht[3 * a + 2] * Hnorm2[6 + b];
      const double ih = 1. / h[8];
      for (int k = 0; k < 9; k++) h[k] *= ih;
      const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
      const double s1 = 1. / fmax(h1n, 2.220446049250313e-16), s2 = 1. / fmax(h2n, 2.220446049250313e-16), s3 = 2. / fmax(h1n + h2n, 2.220446049250313e-16);
      double Hm[9], th[3];
      for (int a = 0; a < 3; a++) { Hm[3 * a] = h[3 * a] * s1; Hm[3 * a + 1] = h[3 * a + 1] * s2; th[a] = h[3 * a + 2] * s3; }
      Hm[2] = Hm[3] * Hm[7] - Hm[6] * Hm[4]; Hm[5] = Hm[6] * Hm[1] - Hm[0] * Hm[7]; Hm[8] = Hm[0] * Hm[4] - Hm[3] * Hm[1];
      double rr[3], Rh[9], Rm[9];
      rodrigues_mat(Hm, rr);
      rodrigues_vec(rr, Rh, nullptr);
      for (int a = 0; a < 3; a++) tt[a] = Rh[3 * a] * Tt[0] + Rh[3 * a + 1] * Tt[1] + Rh[3 * a + 2] * Tt[2] + th[a];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rm[3 * a + b] = Rh[3 * a] * Rt[b] + Rh[3 * a + 1] * Rt[3 + b] + Rh[3 * a + 2] * Rt[6 + b];
      rodrigues_mat(Rm, r);
    }
  } else if (init_ok) {
    init_ok = count >= 6;   // "DLT algorithm needs at least 6 points"
    if (init_ok) {
      double LL[144];
      for (int k = 0; k < 144; k++) LL[k] = 0.0;
      for (int e = lane; e < count; e += 64) {
        const float* P = obj + 3 * inl[e];
        const double x = -(((double)img[2 * inl[e]] - cx) * ifx), y = -(((double)img[2 * inl[e] + 1] - cy) * ify), X = P[0], Y = P[1], Z = P[2];
        const double L0[12] = {X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x}, L1[12] = {0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y};
        for (int j = 0; j < 12; j++) for (int k = j; k < 12; k++) LL[12 * j + k] += L0[j] * L0[k] + L1[j] * L1[k];
      }
      for (int j = 0; j < 12; j++) for (int k = j; k < 12; k++) { LL[12 * j + k] = wave_allsum(LL[12 * j + k]); LL[12 * k + j] = LL[12 * j + k]; }
      if (lane == 0) for (int k = 0; k < 144; k++) wsA[k] = LL[k];
      group_sync();
      if (lane < 16) jacobi_eig_group<12>(wsA, wsV, lane);
      if (lane == 0) {
        int o12[12];
        order_desc<12>(wsA, o12);
        for (int k = 0; k < 12; k++) wsOut[k] = wsV[12 * k + o12[11]];
      }
      group_sync();
      double RRt[12];
      for (int k = 0; k < 12; k++) RRt[k] = wsOut[k];
      double RR[9] = {RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10]};
      if (det3(RR) < 0) { for (int k = 0; k < 12; k++) RRt[k] = -RRt[k]; for (int k = 0; k < 9; k++) RR[k] = -RR[k]; }
      double sc = 0;
      for (int k = 0; k < 9; k++) sc += RR[k] * RR[k];
      sc = sqrt(sc);
      init_ok = fabs(sc) > 2.220446049250313e-16;
      if (init_ok) {
        double Rm[9], nr = 0;
        polar_uvt(RR, Rm);
        for (int k = 0; k < 9; k++) nr += Rm[k] * Rm[k];
        nr = sqrt(nr);
        tt[0] = RRt[3] * (nr / sc); tt[1] = RRt[7] * (nr / sc); tt[2] = RRt[11] * (nr / sc);
        rodrigues_mat(Rm, r);
      }
    }
  }
  // ---- CvLevMarq through update(): J^T J / J^T e / |e| over the inliers by the wavefront, the 6 x 6 step in every lane ----
  if (init_ok) {
    double param[6] = {r[0], r[1], r[2], tt[0], tt[1], tt[2]}, prevParam[6], JtJ[36], JtErr[6];
    double prevErrNorm = 1.7976931348623157e308, errNorm = 0.0;
    int lambdaLg10 = -3, iters = 0, state = 0;   // 0 STARTED, 1 CALC_J, 2 CHECK_ERR, 3 DONE
    // residuals (and normal equations) at `param`
    auto evaluate = [&](bool withJ) -> double {
      double Rp[9], dRdr[27], acc[28];
      rodrigues_vec(param, Rp, withJ ? dRdr : nullptr);
      for (int k = 0; k < 28; k++) acc[k] = 0.0;
      for (int e = lane; e < count; e += 64) {
        const float* P = obj + 3 * inl[e];
        const double X = P[0], Y = P[1], Z = P[2];
        double x = Rp[0] * X + Rp[1] * Y + Rp[2] * Z + param[3], y = Rp[3] * X + Rp[4] * Y + Rp[5] * Z + param[4], z = Rp[6] * X + Rp[7] * Y + Rp[8] * Z + param[5];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const double eu = (x * fx + cx) - (double)img[2 * inl[e]], ev = (y * fy + cy) - (double)img[2 * inl[e] + 1];
        acc[27] += eu * eu + ev * ev;
        if (withJ) {
          double Ju[6], Jv[6];
          const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
          for (int j = 0; j < 3; j++) { Ju[3 + j] = fx * dxdt[j]; Jv[3 + j] = fy * dydt[j]; }
          for (int j = 0; j < 3; j++) {
            const double dx0 = X * dRdr[9 * j] + Y * dRdr[9 * j + 1] + Z * dRdr[9 * j + 2], dy0 = X * dRdr[9 * j + 3] + Y * dRdr[9 * j + 4] + Z * dRdr[9 * j + 5];
            const double dz0 = X * dRdr[9 * j + 6] + Y * dRdr[9 * j + 7] + Z * dRdr[9 * j + 8];
            Ju[j] = fx * (z * (dx0 - x * dz0));
            Jv[j] = fy * (z * (dy0 - y * dz0));
          }
          int q = 0;
          for (int a = 0; a < 6; a++) { for (int b = a; b < 6; b++) acc[q++] += Ju[a] * Ju[b] + Jv[a] * Jv[b]; }
          for (int a = 0; a < 6; a++) acc[21 + a] += Ju[a] * eu + Jv[a] * ev;
        }
      }
      const double e2 = wave_allsum(acc[27]);
      if (withJ) {
        int q = 0;
        for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { const double v = wave_allsum(acc[q++]); JtJ[6 * a + b] = v; JtJ[6 * b + a] = v; }
        for (int a = 0; a < 6; a++) JtErr[a] = wave_allsum(acc[21 + a]);
      }
      return sqrt(e2);
    };
    auto step = [&]() {
      const double lambda = exp(lambdaLg10 * log(10.));
      double A[36], d[6];
      for (int k = 0; k < 36; k++) A[k] = JtJ[k];
      for (int k = 0; k < 6; k++) A[7 * k] *= 1. + lambda;
      // solve(JtJN, JtErr, ., DECOMP_SVD): the system is symmetric positive definite but for degenerate point sets — Cholesky (in
      // registers) gives the SVD's solution to rounding; a failed pivot falls back to the eigen-decomposition (minimum-norm solution)
      if (!chol6_solve(A, JtErr, d)) lstsq_svd<6, 6>(A, JtErr, d);
      for (int k = 0; k < 6; k++) param[k] = prevParam[k] - d[k];
    };
    double curNorm = evaluate(true);    // STARTED: J and err at the initial parameters
    state = 1;
    for (int guard = 0; guard < 1000 && state != 3; guard++) {
      if (state == 1) {
        for (int k = 0; k < 6; k++) prevParam[k] = param[k];
        step();
        if (iters == 0) prevErrNorm = curNorm;
        state = 2;
        curNorm = evaluate(false);
      } else {
        errNorm = curNorm;
        if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) { step(); curNorm = evaluate(false); continue; }
        lambdaLg10 = max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int k = 0; k < 6; k++) { dn += (param[k] - prevParam[k]) * (param[k] - prevParam[k]); pn += prevParam[k] * prevParam[k]; }
        if (++iters >= 20 || sqrt(dn) / sqrt(pn) < 1.1920928955078125e-07) { state = 3; break; }
        prevErrNorm = errNorm;
        state = 1;
        curNorm = evaluate(true);
      }
    }
    for (int k = 0; k < 3; k++) { r[k] = param[k]; tt[k] = param[3 + k]; }
    success = 1;
  }
  if (lane == 0) {
    int* oi = (int*)out;
    oi[0] = count; oi[1] = success; oi[2] = 0; oi[3] = 0;
    double* od = (double*)(out + 16);
    for (int k = 0; k < 3; k++) { od[k] = r[k]; od[3 + k] = tt[k]; }
  }
}

}  // namespace dvs
