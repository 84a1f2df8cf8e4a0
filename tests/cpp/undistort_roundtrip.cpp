// tests/cpp/undistort_roundtrip.cpp — dvslam::undistortImagePoints (include/dvslam/calib3d.hpp) against the forward model it inverts:
// pixels distorted with plumb_bob / rational_polynomial coefficients (OpenCV's projectPoints formulas, calibration.cpp) come back to
// the distortion-free pixels.  No GPU: the helper is host code.  Exit code 0 = ok.
#include <cmath>
#include <cstdio>
#include <vector>
#include "dvslam/calib3d.hpp"

static void distort(double x, double y, const double* k, int nk, double& xd, double& yd) {
  double kk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nk; i++) kk[i] = k[i];
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double cd = (1 + kk[0] * r2 + kk[1] * r4 + kk[4] * r6) / (1 + kk[5] * r2 + kk[6] * r4 + kk[7] * r6);
  xd = x * cd + 2 * kk[2] * x * y + kk[3] * (r2 + 2 * x * x);
  yd = y * cd + kk[2] * (r2 + 2 * y * y) + 2 * kk[3] * x * y;
}

int main() {
  const double K4[4] = {615.2, 613.9, 322.4, 241.7};
  const double D5[5] = {-0.21, 0.05, 0.0011, -0.0007, 0.003};
  const double D8[8] = {0.12, -0.04, 0.0005, 0.0009, 0.002, 0.31, -0.02, 0.001};
  const double D4[4] = {-0.1, 0.02, 0.001, 0.0};
  const struct { const double* d; int n; double tol; } cases[] = {{D5, 5, 0.02}, {D8, 8, 0.02}, {D4, 4, 0.02}, {nullptr, 0, 0.0}};
  for (const auto& c : cases) {
    std::vector<float> ideal, dist;
    for (int v = 20; v < 480; v += 23)
      for (int u = 20; u < 640; u += 31) {
        const double x = (u - K4[2]) / K4[0], y = (v - K4[3]) / K4[1];
        double xd = x, yd = y;
        if (c.n) distort(x, y, c.d, c.n, xd, yd);
        ideal.push_back((float)u); ideal.push_back((float)v);
        dist.push_back((float)(xd * K4[0] + K4[2])); dist.push_back((float)(yd * K4[1] + K4[3]));
      }
    std::vector<float> und;
    dvslam::undistortImagePoints(dist.data(), (int)(dist.size() / 2), K4, c.d, c.n, und);
    double worst = 0, moved = 0;
    for (size_t i = 0; i < und.size(); i++) { worst = std::fmax(worst, std::fabs(und[i] - ideal[i])); moved = std::fmax(moved, std::fabs(dist[i] - ideal[i])); }
    std::printf("%d coefficients: distortion up to %.2f px, residual after undistortion %.4f px\n", c.n, moved, worst);
    if (worst > c.tol || (c.n && moved < 1.0)) return 1;   // five fixed-point passes (OpenCV's default) leave a few 1e-3 px at the corners
  }
  bool refused = false;
  try { std::vector<float> o; const double D12[12] = {0}; const float p[2] = {1, 2}; dvslam::undistortImagePoints(p, 1, K4, D12, 12, o); } catch (const std::invalid_argument&) { refused = true; }
  if (!refused) return 2;
  std::printf("undistort round trip ok\n");
  return 0;
}
