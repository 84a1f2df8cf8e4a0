// pnp_cv_oracle.cpp — CPU checker of cv::solvePnPRansac AS OPENCV 4.x RUNS IT for the reference's call
//   cv::solvePnPRansac(points3d, points2d, K, dist, rvec, tvec, false, 100, 4.0, 0.99, inliers)        frontend.cpp:911-921
// TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline); nothing under dynamic-visual-slam_amd/ may link or call it.
//
// Restated from the published OpenCV 4.x sources (calib3d: solvepnp.cpp, ptsetreg.cpp, epnp.cpp, calibration.cpp; none of them is in
// /root/reference — OpenCV is an un-vendored dependency, SURVEY.md §8c — PARITY UNPINNED):
//   * flags = SOLVEPNP_ITERATIVE, more than 5 points: RANSACPointSetRegistrator(PnPRansacCallback, modelPoints = 5, threshold 4,
//     confidence 0.99, maxIters 100) with ONE cv::RNG seeded (uint64)-1; getSubset = 5 distinct indices (rng.uniform(0, count), an index
//     drawn again while it repeats; the callback has no checkSubset);
//   * runKernel = solvePnP(SOLVEPNP_EPNP) on the 5 points: undistortPoints (no distortion: ((double)u - cx) * (1 / fx), stored as float),
//     class epnp (control points from the PCA of the object points, barycentric coordinates through cvInvert(CV_SVD), M^T M, its four
//     smallest eigenvectors, the three beta approximations each followed by 5 Gauss-Newton steps with epnp::qr_solve, R and t from the
//     3 x 3 SVD of the correlation, the solution with the smallest mean reprojection error), cv::Rodrigues(R);
//   * computeError = projectPoints into CV_32F (z -> 1 / z, then products; pixel stored as float), float squared distance, inlier when
//     err <= (float)(4 * 4); a model replaces the best one when its count exceeds max(best, 4), niters = RANSACUpdateNumIters(0.99,
//     (n - good) / n, 5, niters);
//   * then solvePnP(SOLVEPNP_ITERATIVE) on the inliers (cvFindExtrinsicCameraParams2: planar test on the singular values of the
//     object points' scatter, homography or 12 x 12 DLT initialisation, CvLevMarq with max_iter 20 and epsilon FLT_EPSILON on the
//     analytic projection Jacobian); the inlier list is the RANSAC stage's mask.
// What this statement does NOT reproduce bit for bit: OpenCV's SVD arithmetic (its Jacobi or LAPACK, by the build) — every SVD here is a
// symmetric Jacobi eigen-decomposition, so models agree to rounding and masks wherever no error sits within rounding of the threshold —
// and cv::findHomography's Levenberg-Marquardt polish of the planar initialisation (absorbed by the 20 iterations that follow it).
// Fewer than 6 points are refused (the reference returns before the call, frontend.cpp:900).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct CvRng {   // cv::RNG (MWC), as in ransac_oracle.cpp
  uint64_t state;
  explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffull) {}
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690ull + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

int updateNumIters(double p, double ep, int modelPoints, int maxIters) {   // cv::RANSACUpdateNumIters
  p = std::min(std::max(p, 0.0), 1.0);
  ep = std::min(std::max(ep, 0.0), 1.0);
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, modelPoints);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)std::rint(num / denom);
}

// symmetric eigen-decomposition by cyclic Jacobi; eigenvalues DESCENDING in w, eigenvectors in the ROWS of Vt (= cvSVD's U^T of a
// symmetric positive semi-definite matrix)
void symEigDesc(const double* Ain, int n, double* w, double* Vt) {
  std::vector<double> A(Ain, Ain + (size_t)n * n), V((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 80; sweep++) {
    double off = 0, diag = 0;
    for (int p = 0; p < n; p++) { diag += A[(size_t)p * n + p] * A[(size_t)p * n + p]; for (int q = p + 1; q < n; q++) off += A[(size_t)p * n + q] * A[(size_t)p * n + q]; }
    if (off <= 1e-29 * diag || off == 0) break;   // sum of squares: off-diagonal norm below 3e-15 of the diagonal's (one more sweep squares it)
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) { const double a = A[(size_t)k * n + p], b = A[(size_t)k * n + q]; A[(size_t)k * n + p] = c * a - s * b; A[(size_t)k * n + q] = s * a + c * b; }
        for (int k = 0; k < n; k++) { const double a = A[(size_t)p * n + k], b = A[(size_t)q * n + k]; A[(size_t)p * n + k] = c * a - s * b; A[(size_t)q * n + k] = s * a + c * b; }
        for (int k = 0; k < n; k++) { const double a = V[(size_t)k * n + p], b = V[(size_t)k * n + q]; V[(size_t)k * n + p] = c * a - s * b; V[(size_t)k * n + q] = s * a + c * b; }
      }
  }
  std::vector<int> ord(n);
  for (int i = 0; i < n; i++) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return A[(size_t)a * n + a] > A[(size_t)b * n + b]; });
  for (int i = 0; i < n; i++) {
    w[i] = A[(size_t)ord[i] * n + ord[i]];
    for (int k = 0; k < n; k++) Vt[(size_t)i * n + k] = V[(size_t)k * n + ord[i]];
  }
}

// SVD of a general m x n matrix (m >= n) from the eigen-decomposition of A^T A: w descending, Vt rows, U columns (m x n).  Singular
// values at rounding level get unit vectors completed by Gram-Schmidt (3 x 3 only needs the cross product; general: left zero)
void svdViaEig(const double* A, int m, int n, double* w, double* U, double* Vt) {
  std::vector<double> AtA((size_t)n * n, 0.0), ev(n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < m; k++) s += A[(size_t)k * n + i] * A[(size_t)k * n + j]; AtA[(size_t)i * n + j] = s; }
  symEigDesc(AtA.data(), n, ev.data(), Vt);
  for (int i = 0; i < n; i++) {
    w[i] = std::sqrt(std::max(ev[i], 0.0));
    for (int k = 0; k < m; k++) { double s = 0; for (int j = 0; j < n; j++) s += A[(size_t)k * n + j] * Vt[(size_t)i * n + j]; U[(size_t)k * n + i] = w[i] > 0 ? s / w[i] : 0.0; }
  }
}

// cvSolve(A, b, x, CV_SVD): minimum-norm least squares, singular values <= 2 eps sum(w) dropped (SVD::backSubst)
void solveSvd(const double* A, int m, int n, const double* b, double* x) {
  std::vector<double> w(n), U((size_t)m * n), Vt((size_t)n * n);
  svdViaEig(A, m, n, w.data(), U.data(), Vt.data());
  double thr = 0;
  for (int i = 0; i < n; i++) thr += w[i];
  thr *= 2 * DBL_EPSILON;
  for (int j = 0; j < n; j++) x[j] = 0;
  for (int i = 0; i < n; i++) {
    if (w[i] <= thr) continue;
    double s = 0;
    for (int k = 0; k < m; k++) s += U[(size_t)k * n + i] * b[k];
    s /= w[i];
    for (int j = 0; j < n; j++) x[j] += Vt[(size_t)i * n + j] * s;
  }
}

double det3(const double* R) {
  return R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
}

// U V^T of the SVD of a 3 x 3 matrix (row-major), the third left vector completed by the cross product when the matrix is singular
void polarUVt(const double* A, double* R, double* wOut = nullptr) {
  double w[3], U[9], Vt[9];
  svdViaEig(A, 3, 3, w, U, Vt);
  if (!(w[2] > 1e-14 * w[0])) {   // rank deficient: u3 = u1 x u2 (sign as the SVD of a proper rotation would give; callers fix det)
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * Vt[j] + U[3 * i + 1] * Vt[3 + j] + U[3 * i + 2] * Vt[6 + j];
  if (wOut) { wOut[0] = w[0]; wOut[1] = w[1]; wOut[2] = w[2]; }
}

// cv::Rodrigues, vector -> matrix, with the 3 x 9 Jacobian dR(k) / dr(i) at J[9 i + k] (cvRodrigues2)
void rodriguesVec(const double* rv, double* R, double* J) {
  const double theta = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  if (theta < DBL_EPSILON) {
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(R, I, sizeof(I));
    if (J) { memset(J, 0, 27 * sizeof(double)); J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
    return;
  }
  const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = 1. / theta;
  const double r[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
  const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
  const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
  if (J) {
    const double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0,  0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0,  0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
    const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,  0, 0, 1, 0, 0, 0, -1, 0, 0,  0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
      const double ri = r[i];
      const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri, a4 = s * itheta;
      for (int k = 0; k < 9; k++) J[9 * i + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[9 * i + k] + a3 * rx[k] + a4 * drx[9 * i + k];
    }
  }
}

// cv::Rodrigues, matrix -> vector: the matrix is first replaced by its nearest rotation (U V^T of its SVD)
void rodriguesMat(const double* Rin, double* rv) {
  double R[9];
  polarUVt(Rin, R);
  double r[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double s = std::sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  double theta = std::acos(c);
  if (s < 1e-5) {
    if (c > 0) { rv[0] = rv[1] = rv[2] = 0; return; }
    double t;
    t = (R[0] + 1) * 0.5; r[0] = std::sqrt(std::max(t, 0.));
    t = (R[4] + 1) * 0.5; r[1] = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
    t = (R[8] + 1) * 0.5; r[2] = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
    if (std::fabs(r[0]) < std::fabs(r[1]) && std::fabs(r[0]) < std::fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
    theta /= std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; k++) rv[k] = r[k] * theta;
    return;
  }
  const double vth = 1 / (2 * s) * theta;
  for (int k = 0; k < 3; k++) rv[k] = r[k] * vth;
}

// ---------------------------------------------------------------- class epnp (calib3d/src/epnp.cpp) -------------------------
struct Epnp {
  double uc, vc, fu, fv;
  int n;
  std::vector<double> pws, us, alphas, pcs;
  double cws[4][3], ccs[4][3];

  static double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
  static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
  }

  void choose_control_points() {
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    double m[9] = {0};
    for (int i = 0; i < n; i++)
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) m[3 * a + b] += (pws[3 * i + a] - cws[0][a]) * (pws[3 * i + b] - cws[0][b]);
    double dc[3], uct[9];
    symEigDesc(m, 3, dc, uct);
    for (int i = 1; i < 4; i++) {
      const double k = std::sqrt(std::max(dc[i - 1], 0.0) / n);
      for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }

  void compute_barycentric_coordinates() {
    double cc[9], w[3], U[9], Vt[9], ci[9] = {0};
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    svdViaEig(cc, 3, 3, w, U, Vt);                 // cvInvert(CV_SVD): V diag(1 / w) U^T over the singular values above the threshold
    const double thr = 2 * DBL_EPSILON * (w[0] + w[1] + w[2]);
    for (int k = 0; k < 3; k++) {
      if (w[k] <= thr) continue;
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) ci[3 * a + b] += Vt[3 * k + a] * U[3 * b + k] / w[k];
    }
    for (int i = 0; i < n; i++) {
      const double* pi = &pws[3 * i];
      double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }

  void compute_ccs(const double* betas, const double* ut) {
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
      const double* v = ut + 12 * (11 - i);
      for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
  }
  void compute_pcs() {
    for (int i = 0; i < n; i++) {
      const double* a = &alphas[4 * i];
      double* pc = &pcs[3 * i];
      for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
  }
  void solve_for_sign() {
    if (pcs[2] < 0.0) {
      for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
      for (int i = 0; i < n; i++) { pcs[3 * i] = -pcs[3 * i]; pcs[3 * i + 1] = -pcs[3 * i + 1]; pcs[3 * i + 2] = -pcs[3 * i + 2]; }
    }
  }
  void estimate_R_and_t(double R[3][3], double t[3]) {
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0};
    for (int i = 0; i < n; i++) {
      const double* pc = &pcs[3 * i];
      const double* pw = &pws[3 * i];
      for (int j = 0; j < 3; j++) {
        abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
        abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
        abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
      }
    }
    double Rm[9];
    polarUVt(abt, Rm);                              // R[i][j] = dot(row i of U, row j of V) = (U V^T)[i][j]
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = Rm[3 * i + j];
    if (det3(Rm) < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot(R[0], pw0); t[1] = pc0[1] - dot(R[1], pw0); t[2] = pc0[2] - dot(R[2], pw0);
  }
  double reprojection_error(const double R[3][3], const double t[3]) {
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
      const double* pw = &pws[3 * i];
      const double Xc = dot(R[0], pw) + t[0], Yc = dot(R[1], pw) + t[1], inv_Zc = 1.0 / (dot(R[2], pw) + t[2]);
      const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
      const double u = us[2 * i], v = us[2 * i + 1];
      sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
  }
  double compute_R_and_t(const double* ut, const double* betas, double R[3][3], double t[3]) {
    compute_ccs(betas, ut);
    compute_pcs();
    solve_for_sign();
    estimate_R_and_t(R, t);
    return reprojection_error(R, t);
  }
  void compute_rho(double* rho) {
    rho[0] = dist2(cws[0], cws[1]); rho[1] = dist2(cws[0], cws[2]); rho[2] = dist2(cws[0], cws[3]);
    rho[3] = dist2(cws[1], cws[2]); rho[4] = dist2(cws[1], cws[3]); rho[5] = dist2(cws[2], cws[3]);
  }
  void compute_L_6x10(const double* ut, double* l) {
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
        dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
        dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = l + 10 * i;
      row[0] = dot(dv[0][i], dv[0][i]);
      row[1] = 2.0f * dot(dv[0][i], dv[1][i]);
      row[2] = dot(dv[1][i], dv[1][i]);
      row[3] = 2.0f * dot(dv[0][i], dv[2][i]);
      row[4] = 2.0f * dot(dv[1][i], dv[2][i]);
      row[5] = dot(dv[2][i], dv[2][i]);
      row[6] = 2.0f * dot(dv[0][i], dv[3][i]);
      row[7] = 2.0f * dot(dv[1][i], dv[3][i]);
      row[8] = 2.0f * dot(dv[2][i], dv[3][i]);
      row[9] = dot(dv[3][i], dv[3][i]);
    }
  }
  void find_betas_approx_1(const double* L, const double* rho, double* betas) {   // [B11 B12 B13 B14]
    double l[24], b4[4];
    for (int i = 0; i < 6; i++) { l[4 * i] = L[10 * i]; l[4 * i + 1] = L[10 * i + 1]; l[4 * i + 2] = L[10 * i + 3]; l[4 * i + 3] = L[10 * i + 6]; }
    solveSvd(l, 6, 4, rho, b4);
    if (b4[0] < 0) { betas[0] = std::sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
    else { betas[0] = std::sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0]; }
  }
  void find_betas_approx_2(const double* L, const double* rho, double* betas) {   // [B11 B12 B22]
    double l[18], b3[3];
    for (int i = 0; i < 6; i++) { l[3 * i] = L[10 * i]; l[3 * i + 1] = L[10 * i + 1]; l[3 * i + 2] = L[10 * i + 2]; }
    solveSvd(l, 6, 3, rho, b3);
    if (b3[0] < 0) { betas[0] = std::sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
    else { betas[0] = std::sqrt(b3[0]); betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0; betas[3] = 0.0;
  }
  void find_betas_approx_3(const double* L, const double* rho, double* betas) {   // [B11 B12 B22 B13 B23]
    double l[30], b5[5];
    for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) l[5 * i + k] = L[10 * i + k];
    solveSvd(l, 6, 5, rho, b5);
    if (b5[0] < 0) { betas[0] = std::sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
    else { betas[0] = std::sqrt(b5[0]); betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
  }
  // epnp::qr_solve for the 6 x 4 system, including its pivot scan (which never looks at the last row)
  static void qr_solve(double* pA, double* pb, double* pX, int nr, int nc) {
    double A1[8], A2[8];
    double* ppAkk = pA;
    for (int k = 0; k < nc; k++) {
      double* ppAik1 = ppAkk;
      double eta = std::fabs(*ppAik1);
      for (int i = k + 1; i < nr; i++) { const double elt = std::fabs(*ppAik1); if (eta < elt) eta = elt; ppAik1 += nc; }
      if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
      double* ppAik2 = ppAkk;
      double sum2 = 0.0;
      const double inv_eta = 1. / eta;
      for (int i = k; i < nr; i++) { *ppAik2 *= inv_eta; sum2 += *ppAik2 * *ppAik2; ppAik2 += nc; }
      double sigma = std::sqrt(sum2);
      if (*ppAkk < 0) sigma = -sigma;
      *ppAkk += sigma;
      A1[k] = sigma * *ppAkk;
      A2[k] = -eta * sigma;
      for (int j = k + 1; j < nc; j++) {
        double* ppAik = ppAkk;
        double sum = 0;
        for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
        const double tau = sum / A1[k];
        ppAik = ppAkk;
        for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
      }
      ppAkk += nc + 1;
    }
    double* ppAjj = pA;
    for (int j = 0; j < nc; j++) {
      double* ppAij = ppAjj;
      double tau = 0;
      for (int i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
      tau /= A1[j];
      ppAij = ppAjj;
      for (int i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
      ppAjj += nc + 1;
    }
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
      const double* ppAij = pA + i * nc + (i + 1);
      double sum = 0;
      for (int j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
      pX[i] = (pb[i] - sum) / A2[i];
    }
  }
  void gauss_newton(const double* L, const double* rho, double* betas) {
    for (int k = 0; k < 5; k++) {
      double A[24], b[6], x[4] = {0, 0, 0, 0};
      for (int i = 0; i < 6; i++) {
        const double* rowL = L + i * 10;
        double* rowA = A + i * 4;
        rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
        rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
        rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
        rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
        b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] + rowL[3] * betas[0] * betas[2] +
                         rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] + rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] +
                         rowL[8] * betas[2] * betas[3] + rowL[9] * betas[3] * betas[3]);
      }
      qr_solve(A, b, x, 6, 4);
      for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
  }
  void compute_pose(double R[3][3], double t[3]) {
    choose_control_points();
    compute_barycentric_coordinates();
    std::vector<double> M((size_t)2 * n * 12);
    for (int i = 0; i < n; i++) {
      double* M1 = &M[(size_t)2 * i * 12];
      double* M2 = M1 + 12;
      const double* as = &alphas[4 * i];
      const double u = us[2 * i], v = us[2 * i + 1];
      for (int k = 0; k < 4; k++) {
        M1[3 * k] = as[k] * fu; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (uc - u);
        M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * fv; M2[3 * k + 2] = as[k] * (vc - v);
      }
    }
    double mtm[144], d[12], ut[144];
    for (int a = 0; a < 12; a++) for (int b = 0; b < 12; b++) { double s = 0; for (int k = 0; k < 2 * n; k++) s += M[(size_t)k * 12 + a] * M[(size_t)k * 12 + b]; mtm[12 * a + b] = s; }
    symEigDesc(mtm, 12, d, ut);
    double l_6x10[60], rho[6];
    compute_L_6x10(ut, l_6x10);
    compute_rho(rho);
    double Betas[4][4] = {}, rep_errors[4] = {}, Rs[4][3][3] = {}, ts[4][3] = {};
    find_betas_approx_1(l_6x10, rho, Betas[1]); gauss_newton(l_6x10, rho, Betas[1]); rep_errors[1] = compute_R_and_t(ut, Betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l_6x10, rho, Betas[2]); gauss_newton(l_6x10, rho, Betas[2]); rep_errors[2] = compute_R_and_t(ut, Betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l_6x10, rho, Betas[3]); gauss_newton(l_6x10, rho, Betas[3]); rep_errors[3] = compute_R_and_t(ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep_errors[2] < rep_errors[1]) N = 2;
    if (rep_errors[3] < rep_errors[N]) N = 3;
    memcpy(R, Rs[N], sizeof(Rs[N])); memcpy(t, ts[N], sizeof(ts[N]));
  }
};

// solvePnP(SOLVEPNP_EPNP) on m float correspondences (no distortion) -> rvec, tvec
void solveEpnp(const float* obj, const float* img, const int* idx, int m, const double* K4, double* rvec, double* tvec) {
  Epnp e;
  e.fu = K4[0]; e.fv = K4[1]; e.uc = K4[2]; e.vc = K4[3];
  e.n = m; e.pws.resize(3 * m); e.us.resize(2 * m); e.alphas.resize(4 * m); e.pcs.resize(3 * m);
  const double ifx = 1. / K4[0], ify = 1. / K4[1];
  for (int i = 0; i < m; i++) {
    const int s = idx ? idx[i] : i;
    for (int k = 0; k < 3; k++) e.pws[3 * i + k] = obj[3 * s + k];
    const float xn = (float)(((double)img[2 * s] - K4[2]) * ifx), yn = (float)(((double)img[2 * s + 1] - K4[3]) * ify);   // undistortPoints into CV_32FC2
    e.us[2 * i] = xn * e.fu + e.uc;
    e.us[2 * i + 1] = yn * e.fv + e.vc;
  }
  double R[3][3], t[3];
  e.compute_pose(R, t);
  rodriguesMat(&R[0][0], rvec);
  for (int k = 0; k < 3; k++) tvec[k] = t[k];
}

// PnPRansacCallback::computeError: projectPoints into float, float squared distance
void projErrorsF(const float* obj, const float* img, int n, const double* K4, const double* rvec, const double* tvec, float* err) {
  double R[9];
  rodriguesVec(rvec, R, nullptr);
  for (int i = 0; i < n; i++) {
    const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0], y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1], z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    const float px = (float)(x * K4[0] + K4[2]), py = (float)(y * K4[1] + K4[3]);
    const float dx = img[2 * i] - px, dy = img[2 * i + 1] - py;
    float s = 0;
    s += dx * dx;
    s += dy * dy;
    err[i] = s;
  }
}

// cvProjectPoints2 with dp/dr, dp/dt (no distortion), double in, double out: residual rows 2 i, 2 i + 1
void projectJac(const double* M, int n, const double* rv, const double* tv, const double* K4, double* proj, double* J /* 2n x 6 or null */) {
  double R[9], dRdr[27];
  rodriguesVec(rv, R, dRdr);
  for (int i = 0; i < n; i++) {
    const double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + tv[0], y = R[3] * X + R[4] * Y + R[5] * Z + tv[1], z = R[6] * X + R[7] * Y + R[8] * Z + tv[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    proj[2 * i] = x * K4[0] + K4[2];
    proj[2 * i + 1] = y * K4[1] + K4[3];
    if (!J) continue;
    double* Ju = J + (size_t)(2 * i) * 6;
    double* Jv = Ju + 6;
    const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
    for (int j = 0; j < 3; j++) { Ju[3 + j] = K4[0] * dxdt[j]; Jv[3 + j] = K4[1] * dydt[j]; }
    const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11], X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
    const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14], X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
    const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17], X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
    for (int j = 0; j < 3; j++) {
      Ju[j] = K4[0] * (z * (dx0dr[j] - x * dz0dr[j]));
      Jv[j] = K4[1] * (z * (dy0dr[j] - y * dz0dr[j]));
    }
  }
}

// solve(JtJN, JtErr, x, DECOMP_SVD) for the symmetric 6 x 6 system of CvLevMarq::step
void solveSym6(const double* A, const double* b, double* x) { solveSvd(A, 6, 6, b, x); }

// cvFindExtrinsicCameraParams2(objectPoints (double), imagePoints (double pixels), A, no distortion, useExtrinsicGuess = 0)
bool findExtrinsics(const std::vector<double>& M, const std::vector<double>& m, const double* K4, double* rvec, double* tvec) {
  const int count = (int)M.size() / 3;
  std::vector<double> mn(2 * (size_t)count);
  const double ifx = 1. / K4[0], ify = 1. / K4[1];
  for (int i = 0; i < count; i++) { mn[2 * i] = (m[2 * i] - K4[2]) * ifx; mn[2 * i + 1] = (m[2 * i + 1] - K4[3]) * ify; }
  double Mc[3] = {0, 0, 0};
  for (int i = 0; i < count; i++) for (int k = 0; k < 3; k++) Mc[k] += M[3 * i + k];
  for (int k = 0; k < 3; k++) Mc[k] /= count;
  double MM[9] = {0}, W[3], V[9];
  for (int i = 0; i < count; i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) MM[3 * a + b] += (M[3 * i + a] - Mc[a]) * (M[3 * i + b] - Mc[b]);
  symEigDesc(MM, 3, W, V);     // V = rows (CV_SVD_V_T)
  double r[3], t[3];
  if (W[2] / W[1] < 1e-3) {
    // planar structure: homography between the plane's coordinates and the normalised image
    double Rt[9];
    memcpy(Rt, V, sizeof(V));
    if (V[2] * V[2] + V[5] * V[5] < 1e-10) { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(Rt, I, sizeof(I)); }
    if (det3(Rt) < 0) for (double& v : Rt) v = -v;
    double Tt[3];
    for (int a = 0; a < 3; a++) Tt[a] = -(Rt[3 * a] * Mc[0] + Rt[3 * a + 1] * Mc[1] + Rt[3 * a + 2] * Mc[2]);
    std::vector<double> xy(2 * (size_t)count);
    std::vector<double> mnf(mn);
    for (int i = 0; i < count; i++) {
      xy[2 * i] = Rt[0] * M[3 * i] + Rt[1] * M[3 * i + 1] + Rt[2] * M[3 * i + 2] + Tt[0];
      xy[2 * i + 1] = Rt[3] * M[3 * i] + Rt[4] * M[3 * i + 1] + Rt[5] * M[3 * i + 2] + Tt[1];
      xy[2 * i] = (float)xy[2 * i]; xy[2 * i + 1] = (float)xy[2 * i + 1];          // cv::findHomography works on CV_32F copies of both point sets
      mnf[2 * i] = (float)mn[2 * i]; mnf[2 * i + 1] = (float)mn[2 * i + 1];
    }
    // cv::findHomography(xy, mn, 0): HomographyEstimatorCallback::runKernel (centroid / mean absolute deviation normalisation, the
    // eigenvector of the smallest eigenvalue of L^T L); its Levenberg-Marquardt polish is not restated (see the header)
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    const std::vector<double>& mq = mnf;
    for (int i = 0; i < count; i++) { cm[0] += mq[2 * i]; cm[1] += mq[2 * i + 1]; cM[0] += xy[2 * i]; cM[1] += xy[2 * i + 1]; }
    for (int k = 0; k < 2; k++) { cm[k] /= count; cM[k] /= count; }
    for (int i = 0; i < count; i++) {
      sm[0] += std::fabs(mq[2 * i] - cm[0]); sm[1] += std::fabs(mq[2 * i + 1] - cm[1]);
      sM[0] += std::fabs(xy[2 * i] - cM[0]); sM[1] += std::fabs(xy[2 * i + 1] - cM[1]);
    }
    if (std::fabs(sm[0]) < DBL_EPSILON || std::fabs(sm[1]) < DBL_EPSILON || std::fabs(sM[0]) < DBL_EPSILON || std::fabs(sM[1]) < DBL_EPSILON) return false;
    for (int k = 0; k < 2; k++) { sm[k] = count / sm[k]; sM[k] = count / sM[k]; }
    const double invHnorm[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
    const double Hnorm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double LtL[81] = {0};
    for (int i = 0; i < count; i++) {
      const double x = (mq[2 * i] - cm[0]) * sm[0], y = (mq[2 * i + 1] - cm[1]) * sm[1];
      const double X = (xy[2 * i] - cM[0]) * sM[0], Y = (xy[2 * i + 1] - cM[1]) * sM[1];
      const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
      const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
      for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) LtL[9 * j + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++) for (int k = 0; k < j; k++) LtL[9 * j + k] = LtL[9 * k + j];
    double ew[9], ev[81];
    symEigDesc(LtL, 9, ew, ev);
    const double* H0 = ev + 72;
    double Htemp[9], h[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Htemp[3 * a + b] = invHnorm[3 * a] * H0[b] + invHnorm[3 * a + 1] * H0[3 + b] + invHnorm[3 * a + 2] * H0[6 + b];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) h[3 * a + b] = Htemp[3 * a] * Hnorm2[b] + Htemp[3 * a + 1] * Hnorm2[3 + b] + Htemp[3 * a + 2] * Hnorm2[6 + b];
    for (int k = 0; k < 9; k++) h[k] *= 1. / h[8];
    const double h1n = std::sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), h2n = std::sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
    const double s1 = 1. / std::max(h1n, DBL_EPSILON), s2 = 1. / std::max(h2n, DBL_EPSILON), s3 = 2. / std::max(h1n + h2n, DBL_EPSILON);
    double Hm[9];
    double th[3];
    for (int a = 0; a < 3; a++) { Hm[3 * a] = h[3 * a] * s1; Hm[3 * a + 1] = h[3 * a + 1] * s2; th[a] = h[3 * a + 2] * s3; }
    Hm[2] = Hm[3] * Hm[7] - Hm[6] * Hm[4]; Hm[5] = Hm[6] * Hm[1] - Hm[0] * Hm[7]; Hm[8] = Hm[0] * Hm[4] - Hm[3] * Hm[1];   // h3 = h1 x h2
    double rr[3], Rh[9];
    rodriguesMat(Hm, rr);
    rodriguesVec(rr, Rh, nullptr);
    for (int a = 0; a < 3; a++) t[a] = Rh[3 * a] * Tt[0] + Rh[3 * a + 1] * Tt[1] + Rh[3 * a + 2] * Tt[2] + th[a];
    double Rm[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rm[3 * a + b] = Rh[3 * a] * Rt[b] + Rh[3 * a + 1] * Rt[3 + b] + Rh[3 * a + 2] * Rt[6 + b];
    rodriguesMat(Rm, r);
  } else {
    if (count < 6) return false;
    double LL[144] = {0};
    for (int i = 0; i < count; i++) {
      const double x = -mn[2 * i], y = -mn[2 * i + 1];
      const double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
      const double L0[12] = {X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x};
      const double L1[12] = {0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y};
      for (int j = 0; j < 12; j++) for (int k = 0; k < 12; k++) LL[12 * j + k] += L0[j] * L0[k] + L1[j] * L1[k];
    }
    double LW[12], LV[144];
    symEigDesc(LL, 12, LW, LV);
    double RRt[12];
    memcpy(RRt, LV + 11 * 12, sizeof(RRt));
    double RR[9] = {RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10]};
    if (det3(RR) < 0) { for (double& v : RRt) v = -v; for (double& v : RR) v = -v; }
    double sc = 0;
    for (double v : RR) sc += v * v;
    sc = std::sqrt(sc);
    if (!(std::fabs(sc) > DBL_EPSILON)) return false;
    double Rm[9];
    polarUVt(RR, Rm);
    double nr = 0;
    for (double v : Rm) nr += v * v;
    nr = std::sqrt(nr);
    t[0] = RRt[3] * (nr / sc); t[1] = RRt[7] * (nr / sc); t[2] = RRt[11] * (nr / sc);
    rodriguesMat(Rm, r);
  }
  // CvLevMarq(6, 2 count, (EPS + ITER, 20, FLT_EPSILON), completeSymm) driven through update(): see calibration.cpp
  double param[6] = {r[0], r[1], r[2], t[0], t[1], t[2]}, prevParam[6];
  std::vector<double> J((size_t)2 * count * 6), err(2 * (size_t)count), proj(2 * (size_t)count);
  double JtJ[36], JtErr[6], prevErrNorm = DBL_MAX, errNorm = 0;
  int lambdaLg10 = -3, iters = 0;
  enum { STARTED, CALC_J, CHECK_ERR, DONE } state = STARTED;
  auto step = [&]() {
    const double lambda = std::exp(lambdaLg10 * std::log(10.));
    double A[36];
    memcpy(A, JtJ, sizeof(A));
    for (int k = 0; k < 6; k++) A[7 * k] *= 1. + lambda;
    double d[6];
    solveSym6(A, JtErr, d);
    for (int k = 0; k < 6; k++) param[k] = prevParam[k] - d[k];
  };
  auto residual = [&](bool withJ) {
    projectJac(M.data(), count, param, param + 3, K4, proj.data(), withJ ? J.data() : nullptr);
    for (int k = 0; k < 2 * count; k++) err[k] = proj[k] - m[k];
  };
  auto norm2 = [&]() { double s = 0; for (double v : err) s += v * v; return std::sqrt(s); };
  for (;;) {
    bool needJ = false, proceed = true, needErr = true;
    if (state == STARTED) { needJ = true; state = CALC_J; }
    else if (state == CALC_J) {
      for (int a = 0; a < 6; a++) { JtErr[a] = 0; for (int b = 0; b < 6; b++) JtJ[6 * a + b] = 0; }
      for (int k = 0; k < 2 * count; k++)
        for (int a = 0; a < 6; a++) { JtErr[a] += J[(size_t)k * 6 + a] * err[k]; for (int b = 0; b < 6; b++) JtJ[6 * a + b] += J[(size_t)k * 6 + a] * J[(size_t)k * 6 + b]; }
      memcpy(prevParam, param, sizeof(param));
      step();
      if (iters == 0) prevErrNorm = norm2();
      state = CHECK_ERR;
    } else if (state == CHECK_ERR) {
      errNorm = norm2();
      bool again = false;
      if (errNorm > prevErrNorm) {
        if (++lambdaLg10 <= 16) { step(); again = true; }
      }
      if (!again) {
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int k = 0; k < 6; k++) { dn += (param[k] - prevParam[k]) * (param[k] - prevParam[k]); pn += prevParam[k] * prevParam[k]; }
        if (++iters >= 20 || std::sqrt(dn) / std::sqrt(pn) < FLT_EPSILON) { state = DONE; proceed = true; needErr = false; }   // update() returns true with _err = 0: the caller's loop ends
        else { prevErrNorm = errNorm; needJ = true; state = CALC_J; }
      }
    }
    if (state == DONE || !proceed || !needErr) break;
    residual(needJ);
  }
  for (int k = 0; k < 3; k++) { rvec[k] = param[k]; tvec[k] = param[3 + k]; }
  return true;
}

}  // namespace

extern "C" {

// the RANSAC stage's 5-point subsets (for the index-exact comparison with the product's host-made sequence)
int orc_cv_subsets_nocheck(int n, int modelPoints, int iters, int* idx) {
  CvRng rng((uint64_t)-1);
  for (int it = 0; it < iters; it++)
    for (int i = 0; i < modelPoints; i++) {
      int v;
      for (v = rng.uniform(0, n); std::find(idx + it * modelPoints, idx + it * modelPoints + i, v) != idx + it * modelPoints + i; v = rng.uniform(0, n)) {}
      idx[it * modelPoints + i] = v;
    }
  return iters;
}

// EPnP on the given correspondences (all of them, in order): rvec, tvec
void orc_epnp(const float* obj, const float* img, int m, const double* K4, double* rvec, double* tvec) { solveEpnp(obj, img, nullptr, m, K4, rvec, tvec); }

// solvePnP(SOLVEPNP_ITERATIVE) on double correspondences
int orc_solve_pnp_iterative(const double* obj, const double* img, int n, const double* K4, double* rvec, double* tvec) {
  std::vector<double> M(obj, obj + 3 * (size_t)n), m(img, img + 2 * (size_t)n);
  return findExtrinsics(M, m, K4, rvec, tvec) ? 1 : 0;
}

// returns success.  sel3 = {iteration of the best model, iterations run, its inlier count}; model6 = the RANSAC stage's (rvec, tvec)
int orc_solve_pnp_ransac_cv(const float* obj, const float* img, int n, const double* K4, int iterations, double reprojErr, double confidence,
                            double* rvec, double* tvec, int* inliers, int* nInliers, int* sel3, double* model6) {
  memset(rvec, 0, 24); memset(tvec, 0, 24);
  *nInliers = 0; sel3[0] = -1; sel3[1] = 0; sel3[2] = 0;
  if (n < 6) return 0;
  const int modelPoints = 5;
  const float thr = (float)(reprojErr * reprojErr);
  CvRng rng((uint64_t)-1);
  int niters = std::max(iterations, 1), best = -1, maxGood = 0, it = 0;
  std::vector<float> err(n);
  std::vector<uint8_t> mask(n), bestMask(n, 0);
  double bestModel[6] = {0};
  for (; it < niters; it++) {
    int idx[5];
    for (int i = 0; i < modelPoints; i++) {
      int v;
      for (v = rng.uniform(0, n); std::find(idx, idx + i, v) != idx + i; v = rng.uniform(0, n)) {}
      idx[i] = v;
    }
    double rv[3], tv[3];
    solveEpnp(obj, img, idx, modelPoints, K4, rv, tv);
    projErrorsF(obj, img, n, K4, rv, tv, err.data());
    int good = 0;
    for (int i = 0; i < n; i++) { const int f = err[i] <= thr; mask[i] = (uint8_t)f; good += f; }
    if (good > std::max(maxGood, modelPoints - 1)) {
      std::swap(mask, bestMask);
      memcpy(bestModel, rv, 24); memcpy(bestModel + 3, tv, 24);
      maxGood = good; best = it;
      niters = updateNumIters(confidence, (double)(n - good) / n, modelPoints, niters);
    }
  }
  sel3[0] = best; sel3[1] = it; sel3[2] = maxGood;
  if (model6) memcpy(model6, bestModel, sizeof(bestModel));
  if (maxGood <= 0) return 0;
  std::vector<double> Mi, mi;
  int cnt = 0;
  for (int i = 0; i < n; i++)
    if (bestMask[i]) {
      if (inliers) inliers[cnt] = i;
      cnt++;
      for (int k = 0; k < 3; k++) Mi.push_back(obj[3 * i + k]);
      mi.push_back(img[2 * i]); mi.push_back(img[2 * i + 1]);
    }
  *nInliers = cnt;
  double r[3], t[3];
  if (findExtrinsics(Mi, mi, K4, r, t)) { memcpy(rvec, r, 24); memcpy(tvec, t, 24); return 1; }
  memcpy(rvec, bestModel, 24); memcpy(tvec, bestModel + 3, 24);   // result <= 0: the RANSAC stage's model
  return 0;
}

}  // extern "C"
