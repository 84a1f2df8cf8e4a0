// Compiles the three header-only adapters against the C-ABI and, on a GPU box, runs them on a tiny
// procedurally generated input.  Exit code 0 = ok, 3 = no GPU (expected in the build container).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "dvslam/bf_matcher.hpp"
#include "dvslam/orb_extractor.hpp"
#include "dvslam/sliding_window_ba.hpp"

int main() {
  if (dvs_device_count() < 1) { std::printf("no device: %s\n", "adapters compiled, nothing run"); return 3; }
  const int rows = 480, cols = 640;
  std::vector<uint8_t> img((size_t)rows * cols);
  uint32_t s = 12345;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      s = s * 1664525u + 1013904223u;
      const int base = (((x / 37) + (y / 29)) & 1) ? 190 : 60;
      img[(size_t)y * cols + x] = (uint8_t)(base + (int)((s >> 24) % 17) - 8);
    }
  dvslam::OrbExtractor orb(500, 1.2f, 8, 20, 7);
  std::vector<dvs_keypoint> kps; std::vector<uint8_t> desc;
  const int n = orb(img.data(), rows, cols, cols, kps, desc);
  if (n <= 0) { std::printf("extract failed: %d\n", n); return 1; }
  if (orb(nullptr, 0, 0, 0, kps, desc) != -1) { std::printf("empty image must return -1\n"); return 1; }
  orb(img.data(), rows, cols, cols, kps, desc);
  dvslam::BFMatcher bf;
  std::vector<dvslam::DMatch> m;
  bf.match(desc.data(), n, desc.data(), n, m);
  for (int i = 0; i < n; i++)
    if (m[i].distance != 0.f || m[i].queryIdx != i) { std::printf("self-match %d: idx %d dist %f\n", i, m[i].trainIdx, m[i].distance); return 1; }
  // 2 keyframes, 30 landmarks on a plane, exact observations -> cost ~ 0, converges immediately
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t0[3] = {0, 0, 0}, t1[3] = {0.3, 0, 0};
  std::vector<dvslam::KeyframeData> kf{{7, I, t0}, {9, I, t1}};
  std::vector<dvslam::Landmark> lms; std::vector<dvslam::Observation> obs;
  for (int i = 0; i < 30; i++) {
    const double X = -1.0 + 0.07 * i, Y = 0.5 * std::sin(0.7 * i), Z = 3.0 + 0.05 * i;
    lms.emplace_back(100 + i, "unlabeled", X, Y, Z);
    obs.emplace_back(900 * X / Z + 640, 900 * Y / Z + 360, 100 + i, "unlabeled", 7);
    obs.emplace_back(900 * (X - 0.3) / Z + 640, 900 * Y / Z + 360, 100 + i, "unlabeled", 9);
  }
  dvslam::SlidingWindowBA ba(900, 900, 640, 360);
  dvslam::OptimizationResult r = ba.optimize(kf, lms, obs, 10);
  std::printf("orb %d keypoints; BA success=%d cost=%.3e msg=%s\n", n, (int)r.success, r.final_cost, r.message.c_str());
  if (!r.success || r.final_cost > 1e-12 || r.optimized_poses.size() != 2 || r.optimized_landmarks.size() != 30) return 1;
  if (ba.last_linear_solver() != 1) { std::printf("a 2-keyframe window must run on the device solver, got %d\n", ba.last_linear_solver()); return 1; }
  // a window beyond the device solver's 16 free keyframes (the reference accepts any window, bundle_adjustment.hpp:737-898): it still
  // optimises — normal equations on the host — and says so: once on stderr and in last_linear_solver()
  std::vector<dvslam::KeyframeData> kf20;
  std::vector<dvslam::Observation> obs20;
  for (int c = 0; c < 20; c++) {
    const double tc[3] = {-0.05 * c, 0, 0};   // world -> camera translation of a camera at x = 0.05 c (the smoke test's convention: I, t)
    kf20.emplace_back(100 + c, I, tc);
    for (int i = 0; i < 30; i++) {
      const double X = -1.0 + 0.07 * i, Y = 0.5 * std::sin(0.7 * i), Z = 3.0 + 0.05 * i;
      obs20.emplace_back(900 * (X - 0.05 * c) / Z + 640 + 0.3 * std::sin(1.3 * i + c), 900 * Y / Z + 360 + 0.3 * std::cos(0.9 * i + 2 * c), 100 + i, "unlabeled", 100 + c);
    }
  }
  dvslam::OptimizationResult r20 = ba.optimize(kf20, lms, obs20, 10);
  std::printf("BA 20 keyframes: success=%d cost=%.3e solver=%d msg=%s\n", (int)r20.success, r20.final_cost, ba.last_linear_solver(), r20.message.c_str());
  if (r20.optimized_poses.size() != 20 || ba.last_linear_solver() != 2 || !(r20.final_cost < 30.0)) return 1;
  return 0;
}
