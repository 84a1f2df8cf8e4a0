// ============================================================================================
// oracle/ba_oracle.cpp — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the reference's sliding-window bundle-adjustment evaluation and solve:
//   WeightedSquaredReprojectionError::operator()      bundle_adjustment.hpp:531-565  (restated literally, templated)
//   ceres::AutoDiffCostFunction<..., 2, 4, 3, 3>       bundle_adjustment.hpp:589-592  -> forward-mode Jet<double,10> below
//   ceres::QuaternionRotatePoint                       bundle_adjustment.hpp:537      -> Ceres 2.x rotation.h algorithm
//   ceres::HuberLoss(1.345) + Corrector                bundle_adjustment.hpp:818      -> loss_function.cc / corrector.cc
//   ceres::EigenQuaternionManifold                     bundle_adjustment.hpp:777      -> manifold.cc Plus / PlusJacobian,
//                                                      applied to the (w,x,y,z) memory exactly as the reference does
//   ceres::Solve (LM, SPARSE_SCHUR, options :839-847)  bundle_adjustment.hpp:850-851  -> trust_region_minimizer.cc +
//                                                      levenberg_marquardt_strategy.cc schedule, dense Schur complement
// Ceres (>= 2.1, un-pinned by CMakeLists.txt:20; Ubuntu 24.04 ships 2.2.0) and Eigen are absent
// from the image, so nothing here could be compared with the real libraries: PARITY UNPINNED.  The
// restatement is anchored on the functor text in the reference and Ceres' published algorithms;
// structural known-answers (zero residual/Jacobian behind the camera, cost 0 fixed point with
// noise-free data, Jet derivative == finite differences) are asserted in tests/test_oracle_ba.py.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
// ============================================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ---- ceres::Jet<double, N> (jet.h): value + N partials, the operators the functor needs -------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0) { for (int i = 0; i < N; i++) v[i] = 0; }
  explicit Jet(double s) : a(s) { for (int i = 0; i < N; i++) v[i] = 0; }
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; i++) v[i] = 0; v[k] = 1.0; }
};
template <int N> Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; i++) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; i++) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; i++) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  // jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; (f.v - f_a_by_g_a * g.v) * g_a_inverse
  Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < N; i++) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h;
}
template <int N> Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> Jet<N> sqrt(const Jet<N>& f) { Jet<N> h; h.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * h.a); for (int i = 0; i < N; i++) h.v[i] = f.v[i] * t; return h; }
template <int N> bool operator<=(const Jet<N>& f, const Jet<N>& g) { return f.a <= g.a; }
inline double sqrt(double x) { return std::sqrt(x); }

// ---- ceres::QuaternionRotatePoint / UnitQuaternionRotatePoint (rotation.h, Ceres 2.x) ---------
template <typename T>
void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  T uv0 = q[2] * pt[2] - q[3] * pt[1];
  T uv1 = q[3] * pt[0] - q[1] * pt[2];
  T uv2 = q[1] * pt[1] - q[2] * pt[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  result[0] = pt[0] + q[0] * uv0;
  result[1] = pt[1] + q[0] * uv1;
  result[2] = pt[2] + q[0] * uv2;
  result[0] += q[2] * uv2 - q[3] * uv1;
  result[1] += q[3] * uv0 - q[1] * uv2;
  result[2] += q[1] * uv1 - q[2] * uv0;
}
template <typename T>
void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  const T scale = T(1) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
  UnitQuaternionRotatePoint(unit, pt, result);
}

// ---- the reference functor, bundle_adjustment.hpp:469-565 ---------------------------------------
struct WeightedSquaredReprojectionError {
  double observed_x, observed_y, fx, fy, cx, cy, inv_sigma;
  WeightedSquaredReprojectionError(double ox, double oy, double fx_, double fy_, double cx_, double cy_, double sigma_pixels)
      : observed_x(ox), observed_y(oy), fx(fx_), fy(fy_), cx(cx_), cy(cy_), inv_sigma(1.0 / sigma_pixels) {}
  template <typename T>
  bool operator()(const T* const camera_rotation, const T* const camera_translation, const T* const point, T* residuals) const {
    T point_camera[3];
    QuaternionRotatePoint(camera_rotation, point, point_camera);
    point_camera[0] += camera_translation[0];
    point_camera[1] += camera_translation[1];
    point_camera[2] += camera_translation[2];
    if (point_camera[2] <= T(0.1)) {
      residuals[0] = T(0.0);
      residuals[1] = T(0.0);
      return true;
    }
    T predicted_x = T(fx) * point_camera[0] / point_camera[2] + T(cx);
    T predicted_y = T(fy) * point_camera[1] / point_camera[2] + T(cy);
    T error_x = predicted_x - T(observed_x);
    T error_y = predicted_y - T(observed_y);
    residuals[0] = T(inv_sigma) * error_x;
    residuals[1] = T(inv_sigma) * error_y;
    return true;
  }
};

// AutoDiffCostFunction<F,2,4,3,3>::Evaluate: residuals[2], jacobians row-major 2x4, 2x3, 2x3
void autodiffEvaluate(const WeightedSquaredReprojectionError& f, const double* q, const double* t, const double* X,
                      double* r, double* Jq, double* Jt, double* JX) {
  typedef Jet<10> J;
  J jq[4], jt[3], jX[3], res[2];
  for (int i = 0; i < 4; i++) jq[i] = J(q[i], i);
  for (int i = 0; i < 3; i++) jt[i] = J(t[i], 4 + i);
  for (int i = 0; i < 3; i++) jX[i] = J(X[i], 7 + i);
  f(jq, jt, jX, res);
  for (int k = 0; k < 2; k++) {
    r[k] = res[k].a;
    for (int i = 0; i < 4; i++) Jq[k * 4 + i] = res[k].v[i];
    for (int i = 0; i < 3; i++) Jt[k * 3 + i] = res[k].v[4 + i];
    for (int i = 0; i < 3; i++) JX[k * 3 + i] = res[k].v[7 + i];
  }
}

// ---- ceres::HuberLoss::Evaluate (loss_function.cc) ----------------------------------------------
void huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

// ---- ceres::EigenQuaternionManifold (manifold.cc); x is raw memory read as (x,y,z,w) -------------
void eigenQuatPlusJacobian(const double* x, double* jac /*4x3 row-major*/) {
  jac[0] = x[3];  jac[1] = x[2];   jac[2] = -x[1];
  jac[3] = -x[2]; jac[4] = x[3];   jac[5] = x[0];
  jac[6] = x[1];  jac[7] = -x[0];  jac[8] = x[3];
  jac[9] = -x[0]; jac[10] = -x[1]; jac[11] = -x[2];
}
void eigenQuatPlus(const double* x, const double* delta, double* x_plus_delta) {
  const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (norm_delta == 0.0) { for (int i = 0; i < 4; i++) x_plus_delta[i] = x[i]; return; }
  // q_delta = (sin|d|/|d| * d, cos|d|) in Eigen (x,y,z,w) storage; x_plus_delta = q_delta * x  (Eigen quaternion product)
  const double s = std::sin(norm_delta) / norm_delta;
  const double dx = s * delta[0], dy = s * delta[1], dz = s * delta[2], dw = std::cos(norm_delta);
  const double ax = x[0], ay = x[1], az = x[2], aw = x[3];
  // Eigen: (a*b).w = a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z ; .x = a.w*b.x + a.x*b.w + a.y*b.z - a.z*b.y ; ...
  x_plus_delta[3] = dw * aw - dx * ax - dy * ay - dz * az;
  x_plus_delta[0] = dw * ax + dx * aw + dy * az - dz * ay;
  x_plus_delta[1] = dw * ay + dy * aw + dz * ax - dx * az;
  x_plus_delta[2] = dw * az + dz * aw + dx * ay - dy * ax;
}

struct Problem {
  int K = 0, L = 0, R = 0;
  std::vector<double> q, t, X, uv;
  std::vector<int> cam, lm;
  std::vector<uint8_t> pose_fixed, lm_fixed;
  double fx, fy, cx, cy, sigma, huber_a;
  // per trust-region iteration of the last solve: radius used, kind (0 invalid step, 1 accepted, 2 rejected, 3 parameter
  // tolerance reached, 4 function tolerance reached), cost change, model cost change, relative decrease, candidate cost
  std::vector<double> trace;
};

// one residual block as ceres::ResidualBlock::Evaluate delivers it to the program evaluator:
// local-parameterised, loss-corrected.  Jp = 2x6 (rotation tangent, translation), Jl = 2x3.
double evalBlock(const Problem& P, const double* q, const double* t, const double* X, const double* uv,
                 double* r, double* Jp, double* Jl, double* rawr = nullptr, double* rawJq = nullptr, double* rawJt = nullptr,
                 double* rawJX = nullptr) {
  WeightedSquaredReprojectionError f(uv[0], uv[1], P.fx, P.fy, P.cx, P.cy, P.sigma);
  double rr[2], Jq[8], Jt[6], JX[6];
  autodiffEvaluate(f, q, t, X, rr, Jq, Jt, JX);
  if (rawr) { memcpy(rawr, rr, sizeof(rr)); }
  if (rawJq) memcpy(rawJq, Jq, sizeof(Jq));
  if (rawJt) memcpy(rawJt, Jt, sizeof(Jt));
  if (rawJX) memcpy(rawJX, JX, sizeof(JX));
  double pj[12];
  eigenQuatPlusJacobian(q, pj);
  for (int k = 0; k < 2; k++) {
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int i = 0; i < 4; i++) s += Jq[k * 4 + i] * pj[i * 3 + c];
      Jp[k * 6 + c] = s;
      Jp[k * 6 + 3 + c] = Jt[k * 3 + c];
      Jl[k * 3 + c] = JX[k * 3 + c];
    }
  }
  const double sq = rr[0] * rr[0] + rr[1] * rr[1];
  double rho[3];
  huber(P.huber_a, sq, rho);
  const double sc = std::sqrt(rho[1]);  // Corrector with rho[2] <= 0: residual_scaling_ = sqrt_rho1_, alpha = 0
  r[0] = rr[0] * sc; r[1] = rr[1] * sc;
  for (int i = 0; i < 12; i++) Jp[i] *= sc;
  for (int i = 0; i < 6; i++) Jl[i] *= sc;
  return 0.5 * rho[0];
}

// cost-only evaluation: Ceres calls the functor with T = double when no Jacobian is requested (candidate cost of a
// trust-region step), so a / b is a true division here while the Jet path computes a * (1 / b).
double totalCost(const Problem& P, const std::vector<double>& q, const std::vector<double>& t, const std::vector<double>& X) {
  double c = 0;
  for (int i = 0; i < P.R; i++) {
    WeightedSquaredReprojectionError f(P.uv[2 * i], P.uv[2 * i + 1], P.fx, P.fy, P.cx, P.cy, P.sigma);
    double r[2], rho[3];
    f(&q[4 * P.cam[i]], &t[3 * P.cam[i]], &X[3 * P.lm[i]], r);
    huber(P.huber_a, r[0] * r[0] + r[1] * r[1], rho);
    c += 0.5 * rho[0];
  }
  return c;
}

struct Normal {
  std::vector<double> Hpp, Hll, W, g;  // K*36, L*9, R*18, 6K+3L
  double cost = 0;
};
void buildNormal(const Problem& P, const std::vector<double>& q, const std::vector<double>& t, const std::vector<double>& X,
                 Normal& Nn, std::vector<double>* res = nullptr, std::vector<double>* JP = nullptr, std::vector<double>* JL = nullptr) {
  Nn.Hpp.assign((size_t)P.K * 36, 0); Nn.Hll.assign((size_t)P.L * 9, 0); Nn.W.assign((size_t)P.R * 18, 0);
  Nn.g.assign((size_t)6 * P.K + 3 * P.L, 0); Nn.cost = 0;
  if (res) res->assign((size_t)P.R * 2, 0);
  if (JP) JP->assign((size_t)P.R * 12, 0);
  if (JL) JL->assign((size_t)P.R * 6, 0);
  for (int i = 0; i < P.R; i++) {
    const int c = P.cam[i], l = P.lm[i];
    double r[2], Jp[12], Jl[6];
    Nn.cost += evalBlock(P, &q[4 * c], &t[3 * c], &X[3 * l], &P.uv[2 * i], r, Jp, Jl);
    if (res) { (*res)[2 * i] = r[0]; (*res)[2 * i + 1] = r[1]; }
    if (JP) memcpy(&(*JP)[12 * i], Jp, sizeof(Jp));
    if (JL) memcpy(&(*JL)[6 * i], Jl, sizeof(Jl));
    const bool pf = P.pose_fixed[c], lf = P.lm_fixed[l];
    if (!pf) {
      for (int a = 0; a < 6; a++) {
        Nn.g[6 * c + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
        for (int b = 0; b < 6; b++) Nn.Hpp[36 * c + 6 * a + b] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
      }
    }
    if (!lf) {
      for (int a = 0; a < 3; a++) {
        Nn.g[6 * P.K + 3 * l + a] += Jl[a] * r[0] + Jl[3 + a] * r[1];
        for (int b = 0; b < 3; b++) Nn.Hll[9 * l + 3 * a + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
      }
    }
    if (!pf && !lf)
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 3; b++) Nn.W[18 * i + 3 * a + b] = Jp[a] * Jl[b] + Jp[6 + a] * Jl[3 + b];
  }
}

// dense symmetric positive definite solve (Cholesky), returns false if not SPD
bool cholSolve(std::vector<double>& A, int n, std::vector<double>& b) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
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
  const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (det == 0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  B[0] = (e * i - f * h) * id; B[1] = (c * h - b * i) * id; B[2] = (b * f - c * e) * id;
  B[3] = (f * g - d * i) * id; B[4] = (a * i - c * g) * id; B[5] = (c * d - a * f) * id;
  B[6] = (d * h - e * g) * id; B[7] = (b * g - a * h) * id; B[8] = (a * e - b * d) * id;
  return true;
}

struct Summary { int termination, num_successful_steps, num_iterations, reserved; double initial_cost, final_cost; };

// ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT / (SPARSE_)SCHUR, Solver::Options defaults
// except those set at bundle_adjustment.hpp:839-847.  Follows trust_region_minimizer.cc (2.x):
//   loop { check iteration/gradient/radius limits; compute LM step; invalid-step handling;
//          candidate = Plus(x, step * jacobi_scaling); cost; parameter tolerance; function tolerance;
//          accept (relative_decrease > 1e-3) or reject; LM radius update }
Summary solveLM(Problem& P, int max_iterations, double ftol, double gtol, double ptol) {
  const int K = P.K, L = P.L, R = P.R;
  const int NT = 6 * K + 3 * L;
  Summary S{}; S.termination = 1;
  P.trace.clear();
  auto log = [&](double radius_, int kind, double dc, double dm, double rel, double cand) {
    const double row[6] = {radius_, (double)kind, dc, dm, rel, cand};
    P.trace.insert(P.trace.end(), row, row + 6);
  };
  std::vector<double> q = P.q, t = P.t, X = P.X;
  // variable (non-constant, referenced) tangent columns
  std::vector<uint8_t> lmUsed(L, 0), camUsed(K, 0);
  for (int i = 0; i < R; i++) { lmUsed[P.lm[i]] = 1; camUsed[P.cam[i]] = 1; }
  std::vector<uint8_t> active(NT, 0);
  for (int c = 0; c < K; c++) if (!P.pose_fixed[c] && camUsed[c]) for (int a = 0; a < 6; a++) active[6 * c + a] = 1;
  for (int l = 0; l < L; l++) if (!P.lm_fixed[l] && lmUsed[l]) for (int a = 0; a < 3; a++) active[6 * K + 3 * l + a] = 1;

  Normal Nn;
  buildNormal(P, q, t, X, Nn);
  double x_cost = Nn.cost;
  S.initial_cost = x_cost;
  double min_cost = x_cost;
  // Jacobi scaling from the initial Jacobian: 1 / (1 + sqrt(squared column norm))
  std::vector<double> scale(NT, 1.0);
  for (int c = 0; c < K; c++) for (int a = 0; a < 6; a++) scale[6 * c + a] = 1.0 / (1.0 + std::sqrt(Nn.Hpp[36 * c + 7 * a]));
  for (int l = 0; l < L; l++) for (int a = 0; a < 3; a++) scale[6 * K + 3 * l + a] = 1.0 / (1.0 + std::sqrt(Nn.Hll[9 * l + 4 * a]));

  auto xnorm = [&]() {
    double s = 0;
    for (int c = 0; c < K; c++) if (active[6 * c]) { for (int i = 0; i < 4; i++) s += q[4 * c + i] * q[4 * c + i]; for (int i = 0; i < 3; i++) s += t[3 * c + i] * t[3 * c + i]; }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) s += X[3 * l + i] * X[3 * l + i];
    return std::sqrt(s);
  };
  auto gradMaxNorm = [&]() {  // || x - Plus(x, -g) ||_inf
    double m = 0;
    for (int c = 0; c < K; c++) if (active[6 * c]) {
      double d[3] = {-Nn.g[6 * c], -Nn.g[6 * c + 1], -Nn.g[6 * c + 2]}, qp[4];
      eigenQuatPlus(&q[4 * c], d, qp);
      for (int i = 0; i < 4; i++) m = std::max(m, std::fabs(q[4 * c + i] - qp[i]));
      for (int i = 0; i < 3; i++) m = std::max(m, std::fabs(Nn.g[6 * c + 3 + i]));
    }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) m = std::max(m, std::fabs(Nn.g[6 * K + 3 * l + i]));
    return m;
  };

  double radius = 1e4, decrease_factor = 2.0;
  const double max_radius = 1e16, min_radius = 1e-32, min_diag = 1e-6, max_diag = 1e32, min_rel_decrease = 1e-3;
  bool reuse_diagonal = false;
  std::vector<double> diagonal(NT, 0.0);
  int iteration = 0, invalid = 0;
  double grad_max = gradMaxNorm();
  std::vector<int> camSlot(K, -1);
  int nc = 0;
  for (int c = 0; c < K; c++) if (active[6 * c]) camSlot[c] = nc++;
  const int n = 6 * nc;

  while (true) {
    if (iteration >= max_iterations) { S.termination = 1; break; }
    if (grad_max <= gtol) { S.termination = 0; break; }
    if (radius < min_radius) { S.termination = 0; break; }
    iteration++;
    // ---- LM step on the Jacobi-scaled system ----
    if (!reuse_diagonal) {
      for (int c = 0; c < K; c++) for (int a = 0; a < 6; a++) diagonal[6 * c + a] = std::min(std::max(Nn.Hpp[36 * c + 7 * a] * scale[6 * c + a] * scale[6 * c + a], min_diag), max_diag);
      for (int l = 0; l < L; l++) for (int a = 0; a < 3; a++) { const int j = 6 * K + 3 * l + a; diagonal[j] = std::min(std::max(Nn.Hll[9 * l + 4 * a] * scale[j] * scale[j], min_diag), max_diag); }
    }
    reuse_diagonal = true;
    // Schur complement: eliminate landmarks
    std::vector<double> Sm((size_t)n * n, 0.0), rhs(n, 0.0), step(NT, 0.0);
    for (int c = 0; c < K; c++) if (camSlot[c] >= 0) {
      const int o = 6 * camSlot[c];
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) Sm[(size_t)(o + a) * n + o + b] = Nn.Hpp[36 * c + 6 * a + b] * scale[6 * c + a] * scale[6 * c + b];
        Sm[(size_t)(o + a) * n + o + a] += diagonal[6 * c + a] / radius;
        rhs[o + a] = Nn.g[6 * c + a] * scale[6 * c + a];
      }
    }
    std::vector<std::vector<int>> obsOf(L);
    for (int i = 0; i < R; i++) obsOf[P.lm[i]].push_back(i);
    std::vector<double> Vinv((size_t)L * 9, 0.0);
    bool ok = true;
    for (int l = 0; l < L && ok; l++) {
      const int j0 = 6 * K + 3 * l;
      if (!active[j0]) continue;
      double V[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) V[3 * a + b] = Nn.Hll[9 * l + 3 * a + b] * scale[j0 + a] * scale[j0 + b];
      for (int a = 0; a < 3; a++) V[4 * a] += diagonal[j0 + a] / radius;
      if (!inv3(V, &Vinv[9 * l])) { ok = false; break; }
      double gl[3] = {Nn.g[j0] * scale[j0], Nn.g[j0 + 1] * scale[j0 + 1], Nn.g[j0 + 2] * scale[j0 + 2]};
      for (int i : obsOf[l]) {
        const int ci = P.cam[i];
        if (camSlot[ci] < 0) continue;
        double Wi[18], Y[18];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) Wi[3 * a + b] = Nn.W[18 * i + 3 * a + b] * scale[6 * ci + a] * scale[j0 + b];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) Y[3 * a + b] = Wi[3 * a] * Vinv[9 * l + b] + Wi[3 * a + 1] * Vinv[9 * l + 3 + b] + Wi[3 * a + 2] * Vinv[9 * l + 6 + b];
        for (int a = 0; a < 6; a++) rhs[6 * camSlot[ci] + a] -= Y[3 * a] * gl[0] + Y[3 * a + 1] * gl[1] + Y[3 * a + 2] * gl[2];
        for (int k : obsOf[l]) {
          const int ck = P.cam[k];
          if (camSlot[ck] < 0) continue;
          for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            double s = 0;
            for (int m = 0; m < 3; m++) s += Y[3 * a + m] * Nn.W[18 * k + 3 * b + m] * scale[6 * ck + b] * scale[j0 + m];
            Sm[(size_t)(6 * camSlot[ci] + a) * n + 6 * camSlot[ck] + b] -= s;
          }
        }
      }
    }
    if (ok && n > 0) ok = cholSolve(Sm, n, rhs);
    bool step_valid = ok;
    if (ok) {
      for (int c = 0; c < K; c++) if (camSlot[c] >= 0) for (int a = 0; a < 6; a++) step[6 * c + a] = rhs[6 * camSlot[c] + a];
      for (int l = 0; l < L; l++) {
        const int j0 = 6 * K + 3 * l;
        if (!active[j0]) continue;
        double b[3] = {Nn.g[j0] * scale[j0], Nn.g[j0 + 1] * scale[j0 + 1], Nn.g[j0 + 2] * scale[j0 + 2]};
        for (int i : obsOf[l]) {
          const int ci = P.cam[i];
          if (camSlot[ci] < 0) continue;
          for (int m = 0; m < 3; m++) for (int a = 0; a < 6; a++) b[m] -= Nn.W[18 * i + 3 * a + m] * scale[6 * ci + a] * scale[j0 + m] * step[6 * ci + a];
        }
        for (int a = 0; a < 3; a++) step[j0 + a] = Vinv[9 * l + 3 * a] * b[0] + Vinv[9 * l + 3 * a + 1] * b[1] + Vinv[9 * l + 3 * a + 2] * b[2];
      }
      for (int j = 0; j < NT; j++) { step[j] = -step[j]; if (!std::isfinite(step[j])) step_valid = false; }
    }
    double model_cost_change = 0;
    if (step_valid) {
      // -(J s)'(r + J s / 2) = -(s'g~ + s'H~s/2)
      double sg = 0, sHs = 0;
      for (int j = 0; j < NT; j++) sg += step[j] * Nn.g[j] * scale[j];
      for (int c = 0; c < K; c++) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) sHs += step[6 * c + a] * scale[6 * c + a] * Nn.Hpp[36 * c + 6 * a + b] * scale[6 * c + b] * step[6 * c + b];
      for (int l = 0; l < L; l++) { const int j0 = 6 * K + 3 * l; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sHs += step[j0 + a] * scale[j0 + a] * Nn.Hll[9 * l + 3 * a + b] * scale[j0 + b] * step[j0 + b]; }
      for (int i = 0; i < R; i++) { const int c = P.cam[i], j0 = 6 * K + 3 * P.lm[i]; for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) sHs += 2.0 * step[6 * c + a] * scale[6 * c + a] * Nn.W[18 * i + 3 * a + b] * scale[j0 + b] * step[j0 + b]; }
      model_cost_change = -(sg + 0.5 * sHs);
      if (model_cost_change <= 0.0) step_valid = false;
    }
    if (!step_valid) {
      log(radius, 0, 0, model_cost_change, 0, 0);
      if (++invalid >= 5) { S.termination = 2; break; }
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = false;  // StepIsInvalid
      continue;
    }
    invalid = 0;
    // candidate
    std::vector<double> cq = q, ct = t, cX = X;
    for (int c = 0; c < K; c++) if (active[6 * c]) {
      double d[3] = {step[6 * c] * scale[6 * c], step[6 * c + 1] * scale[6 * c + 1], step[6 * c + 2] * scale[6 * c + 2]};
      eigenQuatPlus(&q[4 * c], d, &cq[4 * c]);
      for (int i = 0; i < 3; i++) ct[3 * c + i] = t[3 * c + i] + step[6 * c + 3 + i] * scale[6 * c + 3 + i];
    }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) cX[3 * l + i] = X[3 * l + i] + step[6 * K + 3 * l + i] * scale[6 * K + 3 * l + i];
    const double cand_cost = totalCost(P, cq, ct, cX);
    // parameter tolerance
    double sn = 0;
    for (int c = 0; c < K; c++) if (active[6 * c]) { for (int i = 0; i < 4; i++) sn += (q[4 * c + i] - cq[4 * c + i]) * (q[4 * c + i] - cq[4 * c + i]); for (int i = 0; i < 3; i++) sn += (t[3 * c + i] - ct[3 * c + i]) * (t[3 * c + i] - ct[3 * c + i]); }
    for (int l = 0; l < L; l++) if (active[6 * K + 3 * l]) for (int i = 0; i < 3; i++) sn += (X[3 * l + i] - cX[3 * l + i]) * (X[3 * l + i] - cX[3 * l + i]);
    if (std::sqrt(sn) <= ptol * (xnorm() + ptol)) { log(radius, 3, x_cost - cand_cost, model_cost_change, 0, cand_cost); S.termination = 0; break; }
    // function tolerance
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= ftol * x_cost) { log(radius, 4, cost_change, model_cost_change, 0, cand_cost); S.termination = 0; break; }
    const double rel_decrease = cost_change / model_cost_change;
    log(radius, rel_decrease > min_rel_decrease ? 1 : 2, cost_change, model_cost_change, rel_decrease, cand_cost);
    if (rel_decrease > min_rel_decrease) {
      q = cq; t = ct; X = cX;
      buildNormal(P, q, t, X, Nn);
      x_cost = Nn.cost;
      grad_max = gradMaxNorm();
      S.num_successful_steps++;
      min_cost = std::min(min_cost, x_cost);
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_decrease - 1.0, 3));
      radius = std::min(max_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  S.num_iterations = iteration;
  S.final_cost = min_cost;
  P.q = q; P.t = t; P.X = X;
  return S;
}

}  // namespace

extern "C" {

void* orc_ba_create(int K, const double* q, const double* t, int L, const double* X, int R, const int32_t* cam, const int32_t* lm,
                    const double* uv, const uint8_t* pose_fixed, const uint8_t* lm_fixed, double fx, double fy, double cx, double cy,
                    double sigma, double huber) {
  Problem* P = new Problem();
  P->K = K; P->L = L; P->R = R;
  P->q.assign(q, q + 4 * K); P->t.assign(t, t + 3 * K); P->X.assign(X, X + 3 * L); P->uv.assign(uv, uv + 2 * R);
  P->cam.assign(cam, cam + R); P->lm.assign(lm, lm + R);
  P->pose_fixed.assign(K, 0); P->lm_fixed.assign(L, 0);
  if (pose_fixed) P->pose_fixed.assign(pose_fixed, pose_fixed + K);
  if (lm_fixed) P->lm_fixed.assign(lm_fixed, lm_fixed + L);
  P->fx = fx; P->fy = fy; P->cx = cx; P->cy = cy; P->sigma = sigma; P->huber_a = huber;
  return P;
}
void orc_ba_destroy(void* h) { delete (Problem*)h; }

// raw functor + autodiff outputs per observation (what CostFunction::Evaluate returns)
void orc_ba_evaluate_raw(void* h, double* res, double* Jq, double* Jt, double* JX) {
  Problem& P = *(Problem*)h;
  for (int i = 0; i < P.R; i++) {
    WeightedSquaredReprojectionError f(P.uv[2 * i], P.uv[2 * i + 1], P.fx, P.fy, P.cx, P.cy, P.sigma);
    double r[2], a[8], b[6], c[6];
    autodiffEvaluate(f, &P.q[4 * P.cam[i]], &P.t[3 * P.cam[i]], &P.X[3 * P.lm[i]], r, a, b, c);
    if (res) memcpy(res + 2 * i, r, sizeof(r));
    if (Jq) memcpy(Jq + 8 * i, a, sizeof(a));
    if (Jt) memcpy(Jt + 6 * i, b, sizeof(b));
    if (JX) memcpy(JX + 6 * i, c, sizeof(c));
  }
}
// robustified evaluation: cost, corrected residuals, local Jacobians, gradient
void orc_ba_evaluate(void* h, double* cost, double* res, double* Jp, double* Jl, double* grad) {
  Problem& P = *(Problem*)h;
  Normal Nn; std::vector<double> r, jp, jl;
  buildNormal(P, P.q, P.t, P.X, Nn, &r, &jp, &jl);
  if (cost) *cost = Nn.cost;
  if (res) memcpy(res, r.data(), r.size() * 8);
  if (Jp) memcpy(Jp, jp.data(), jp.size() * 8);
  if (Jl) memcpy(Jl, jl.data(), jl.size() * 8);
  if (grad) memcpy(grad, Nn.g.data(), Nn.g.size() * 8);
}
// `reps` evaluations (residuals, local Jacobians, loss correction, cost and the H_pp / H_ll / g sums) with the residual blocks
// split over `nthreads` threads — how Ceres evaluates with options.num_threads = 4 (bundle_adjustment.hpp:842): every thread
// owns a contiguous slice of the residual blocks and its own accumulators, summed at the end.  Timed CPU baseline only.
void orc_ba_evaluate_mt(void* h, int nthreads, int reps, double* cost) {
  Problem& P = *(Problem*)h;
  nthreads = std::max(1, nthreads);
  std::vector<double> res((size_t)P.R * 2), JP((size_t)P.R * 12), JL((size_t)P.R * 6), W((size_t)P.R * 18);
  double total = 0;
  for (int rep = 0; rep < reps; rep++) {
    std::vector<Normal> part(nthreads);
    std::vector<std::thread> th;
    for (int w = 0; w < nthreads; w++)
      th.emplace_back([&, w]() {
        Normal& Nn = part[w];
        Nn.Hpp.assign((size_t)P.K * 36, 0); Nn.Hll.assign((size_t)P.L * 9, 0); Nn.g.assign((size_t)6 * P.K + 3 * P.L, 0); Nn.cost = 0;
        const int i0 = (int)((long)P.R * w / nthreads), i1 = (int)((long)P.R * (w + 1) / nthreads);
        for (int i = i0; i < i1; i++) {
          const int c = P.cam[i], l = P.lm[i];
          double* r = &res[2 * i]; double* Jp = &JP[12 * i]; double* Jl = &JL[6 * i];
          Nn.cost += evalBlock(P, &P.q[4 * c], &P.t[3 * c], &P.X[3 * l], &P.uv[2 * i], r, Jp, Jl);
          const bool pf = P.pose_fixed[c], lf = P.lm_fixed[l];
          if (!pf)
            for (int a = 0; a < 6; a++) {
              Nn.g[6 * c + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
              for (int b = 0; b < 6; b++) Nn.Hpp[36 * c + 6 * a + b] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
            }
          if (!lf)
            for (int a = 0; a < 3; a++) {
              Nn.g[6 * P.K + 3 * l + a] += Jl[a] * r[0] + Jl[3 + a] * r[1];
              for (int b = 0; b < 3; b++) Nn.Hll[9 * l + 3 * a + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
            }
          if (!pf && !lf)
            for (int a = 0; a < 6; a++)
              for (int b = 0; b < 3; b++) W[18 * i + 3 * a + b] = Jp[a] * Jl[b] + Jp[6 + a] * Jl[3 + b];
        }
      });
    for (auto& t : th) t.join();
    total = 0;
    for (int w = 1; w < nthreads; w++) {
      for (size_t k = 0; k < part[0].Hpp.size(); k++) part[0].Hpp[k] += part[w].Hpp[k];
      for (size_t k = 0; k < part[0].Hll.size(); k++) part[0].Hll[k] += part[w].Hll[k];
      for (size_t k = 0; k < part[0].g.size(); k++) part[0].g[k] += part[w].g[k];
    }
    for (int w = 0; w < nthreads; w++) total += part[w].cost;
  }
  if (cost) *cost = total;
}
void orc_ba_normal_equations(void* h, double* Hpp, double* Hll, double* W, double* g, double* cost) {
  Problem& P = *(Problem*)h;
  Normal Nn;
  buildNormal(P, P.q, P.t, P.X, Nn);
  if (Hpp) memcpy(Hpp, Nn.Hpp.data(), Nn.Hpp.size() * 8);
  if (Hll) memcpy(Hll, Nn.Hll.data(), Nn.Hll.size() * 8);
  if (W) memcpy(W, Nn.W.data(), Nn.W.size() * 8);
  if (g) memcpy(g, Nn.g.data(), Nn.g.size() * 8);
  if (cost) *cost = Nn.cost;
}
void orc_ba_solve(void* h, int max_iterations, double ftol, double gtol, double ptol, void* summary) {
  Summary s = solveLM(*(Problem*)h, max_iterations, ftol, gtol, ptol);
  memcpy(summary, &s, sizeof(s));
}
int orc_ba_get_trace(void* h, double* rows, int cap_rows) {
  Problem& P = *(Problem*)h;
  const int n = (int)(P.trace.size() / 6);
  if (rows) memcpy(rows, P.trace.data(), (size_t)std::min(n, cap_rows) * 6 * 8);
  return n;
}
void orc_ba_get_parameters(void* h, double* q, double* t, double* X) {
  Problem& P = *(Problem*)h;
  memcpy(q, P.q.data(), P.q.size() * 8); memcpy(t, P.t.data(), P.t.size() * 8); memcpy(X, P.X.data(), P.X.size() * 8);
}
void orc_ba_quat_plus(const double* x, const double* delta, double* out) { eigenQuatPlus(x, delta, out); }
void orc_ba_huber(double a, double s, double* rho3) { huber(a, s, rho3); }

// Eigen pose conversions on the boundary (CameraPose::fromRt / toRt, bundle_adjustment.hpp:138-212): Eigen 3.4
// Quaterniond(Matrix3d) + normalize(), toRotationMatrix().  R row-major 3x3.
void orc_ba_from_rt(const double* R_wc, const double* t_wc, double* q_wxyz, double* trans) {
  double m[9];  // R_camera_world = R^T
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[3 * i + j] = R_wc[3 * j + i];
  for (int i = 0; i < 3; i++) trans[i] = -(m[3 * i] * t_wc[0] + m[3 * i + 1] * t_wc[1] + m[3 * i + 2] * t_wc[2]);
  double w, x, y, z;
  double tr = m[0] + m[4] + m[8];
  if (tr > 0) {
    tr = std::sqrt(tr + 1.0); w = 0.5 * tr; tr = 0.5 / tr;
    x = (m[7] - m[5]) * tr; y = (m[2] - m[6]) * tr; z = (m[3] - m[1]) * tr;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double v[3];
    tr = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    v[i] = 0.5 * tr; tr = 0.5 / tr;
    w = (m[3 * k + j] - m[3 * j + k]) * tr;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * tr;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * tr;
    x = v[0]; y = v[1]; z = v[2];
  }
  const double nn = std::sqrt(w * w + x * x + y * y + z * z);  // normalize()
  q_wxyz[0] = w / nn; q_wxyz[1] = x / nn; q_wxyz[2] = y / nn; q_wxyz[3] = z / nn;
}
void orc_ba_to_rt(const double* q_wxyz, const double* trans, double* R_wc, double* t_wc) {
  const double w = q_wxyz[0], x = q_wxyz[1], y = q_wxyz[2], z = q_wxyz[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double Rcw[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R_wc[3 * i + j] = Rcw[3 * j + i];
  for (int i = 0; i < 3; i++) t_wc[i] = -(R_wc[3 * i] * trans[0] + R_wc[3 * i + 1] * trans[1] + R_wc[3 * i + 2] * trans[2]);
}
}
