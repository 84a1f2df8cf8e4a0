// Compiles EVERY `DVSLAM_WITH_OPENCV` branch and both drop-in headers (include/dynamic_visual_slam/*.hpp) against the test-only
// stand-ins in tests/cpp/stubs/, written the way the reference's call sites use the types (frontend.cpp:205-220, 1094-1095,
// 1123; backend.cpp:180, 222, 908-973, 1072, 1356-1392).  On a GPU box it also runs them.  Exit 0 = ok, 3 = no GPU.
#include <cmath>
#include <cstdio>
#include <memory>
#include "dynamic_visual_slam/ORBextractor.hpp"
#include "dynamic_visual_slam/bundle_adjustment.hpp"
#include "dvslam/bf_matcher.hpp"
#include "dvslam/calib3d.hpp"
#include "dvslam/cv_orb.hpp"

int main() {
  if (dvs_device_count() < 1) { std::printf("no device: OpenCV-typed adapters compiled, nothing run\n"); return 3; }
  const int rows = 480, cols = 640;
  cv::Mat gray(rows, cols, CV_8UC1);
  uint32_t s = 12345;
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      s = s * 1664525u + 1013904223u;
      gray.at<uint8_t>(y, x) = (uint8_t)(((((x / 37) + (y / 29)) & 1) ? 190 : 60) + (int)((s >> 24) % 17) - 8);
    }
  // frontend.cpp:205-211, 220
  auto orb_extractor_ = std::make_unique<ORB_SLAM3::ORBextractor>(500, 1.2f, 8, 20, 7);
  dvslam::HammingBFMatcher matcher_(cv::NORM_HAMMING);
  std::vector<int> vLappingArea = {0, 0};
  std::vector<cv::KeyPoint> kps;
  cv::Mat descriptors;
  const int n = (*orb_extractor_)(gray, cv::noArray(), kps, descriptors, vLappingArea);   // frontend.cpp:1094-1095
  if (n <= 0 || descriptors.rows != n || descriptors.cols != 32 || (int)kps.size() != n) { std::printf("extract: %d\n", n); return 1; }
  if ((int)orb_extractor_->mvImagePyramid.size() != orb_extractor_->GetLevels()) { std::printf("mvImagePyramid missing\n"); return 1; }
  const std::vector<float> sf = orb_extractor_->GetScaleFactors();
  for (int l = 0; l < orb_extractor_->GetLevels(); l++) {
    const cv::Mat& L = orb_extractor_->mvImagePyramid[l];
    const float inv = 1.0f / sf[l];
    if (L.cols != (int)std::lrintf((float)cols * inv) || L.rows != (int)std::lrintf((float)rows * inv)) { std::printf("level %d size %dx%d\n", l, L.cols, L.rows); return 1; }
  }
  if (std::memcmp(orb_extractor_->mvImagePyramid[0].data, gray.data, (size_t)rows * cols) != 0) { std::printf("level 0 != input\n"); return 1; }
  cv::Mat empty;
  if ((*orb_extractor_)(empty, cv::noArray(), kps, descriptors, vLappingArea) != -1) { std::printf("empty image must return -1\n"); return 1; }
  (*orb_extractor_)(gray, cv::noArray(), kps, descriptors, vLappingArea);
  std::vector<cv::DMatch> all_matches;
  matcher_.match(descriptors, descriptors, all_matches);                                     // frontend.cpp:1123
  for (int i = 0; i < n; i++)
    if (all_matches[i].distance != 0.f || all_matches[i].queryIdx != i) { std::printf("self-match %d\n", i); return 1; }
  dvslam::HammingBFMatcher descriptor_matcher_(cv::NORM_HAMMING, false);                     // backend.cpp:222
  // backend.cpp:908-960: window assembly + optimize, then :967-977 / :1356-1392 result handling
  std::unique_ptr<SlidingWindowBA> bundle_adjuster_ = std::make_unique<SlidingWindowBA>(900.0, 900.0, 640.0, 360.0);
  std::vector<KeyframeData> window_keyframes;
  for (int k = 0; k < 2; k++) {
    cv::Mat R(3, 3, CV_64F), t(3, 1, CV_64F);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R.at<double>(i, j) = i == j ? 1.0 : 0.0; t.at<double>(i) = 0.0; }
    t.at<double>(0) = 0.3 * k;
    window_keyframes.emplace_back(7 + 2 * k, R, t, rclcpp::Time(1000 * k));
  }
  std::vector<Landmark> landmarks; std::vector<Observation> observations;
  for (int i = 0; i < 30; i++) {
    const double X = -1.0 + 0.07 * i, Y = 0.5 * std::sin(0.7 * i), Z = 3.0 + 0.05 * i;
    landmarks.emplace_back(100 + i, "unlabeled", X, Y, Z);
    observations.emplace_back(900 * X / Z + 640, 900 * Y / Z + 360, 100 + i, "unlabeled", 7);
    observations.emplace_back(900 * (X - 0.3) / Z + 640, 900 * Y / Z + 360, 100 + i, "unlabeled", 9);
  }
  OptimizationResult result = bundle_adjuster_->optimize(window_keyframes, landmarks, observations, 20);
  result = bundle_adjuster_->optimize(window_keyframes, landmarks, observations, 20);     // the handle is reused across calls
  if (!result.success || result.final_cost > 1e-12 || result.optimized_poses.size() != 2 || result.optimized_landmarks.size() != 30) {
    std::printf("BA: %s cost %g\n", result.message.c_str(), result.final_cost); return 1;
  }
  for (const auto& [frame_id, pose_pair] : result.optimized_poses) {                       // backend.cpp:1358-1370
    const auto& [R_opt, t_opt] = pose_pair;
    cv::Mat Rc = R_opt.clone(), tc = t_opt.clone();
    if (Rc.rows != 3 || Rc.cols != 3 || tc.rows != 3 || tc.cols != 1 || frame_id < 7) return 1;
  }
  for (const auto& [landmark_key, optimized_pos] : result.optimized_landmarks) {           // backend.cpp:1373-1387
    const auto& [landmark_id, landmark_category] = landmark_key;
    if (landmark_id < 100 || landmark_category != "unlabeled" || !(optimized_pos.z > 0)) return 1;
  }
  CameraPose cp;                                                                            // bundle_adjustment.hpp:92-213 round trip
  cp.fromRt(window_keyframes[1].R, window_keyframes[1].t);
  cv::Mat R2, t2;
  cp.toRt(R2, t2);
  if (std::fabs(t2.at<double>(0) - 0.3) > 1e-12 || std::fabs(R2.at<double>(1, 1) - 1.0) > 1e-12) return 1;
  {  // test_dbow2_integration.cpp:14-19, 33-43 with cv::ORB -> dvslam::ORB: three filled discs on black, create(100), detectAndCompute
    cv::Mat dummy_image_(480, 640, CV_8UC1);
    const int cx[3] = {100, 300, 500}, cy[3] = {100, 200, 300}, rr[3] = {50, 30, 40};
    for (int y = 0; y < 480; y++)
      for (int x = 0; x < 640; x++) {
        uint8_t v = 0;
        for (int k = 0; k < 3; k++) if ((x - cx[k]) * (x - cx[k]) + (y - cy[k]) * (y - cy[k]) <= rr[k] * rr[k]) v = 255;
        dummy_image_.at<uint8_t>(y, x) = v;
      }
    std::shared_ptr<dvslam::ORB> orb_ = dvslam::ORB::create(100);
    std::vector<cv::KeyPoint> keypoints;
    cv::Mat descriptors2;
    orb_->detectAndCompute(dummy_image_, cv::Mat(), keypoints, descriptors2);
    if (!(descriptors2.rows > 0) || descriptors2.cols != 32 || (int)keypoints.size() != descriptors2.rows) {   // the reference test's assertions (:41-42)
      std::printf("dvslam::ORB: %d x %d descriptors\n", descriptors2.rows, descriptors2.cols); return 1;
    }
    std::printf("dvslam::ORB on the disc image: %d descriptors\n", descriptors2.rows);
  }
  {  // frontend.cpp:627-644 with cv::findFundamentalMat -> dvslam::findFundamentalMat(matcher_, ...): a pure translation between two views plus
     // gross outliers; the mask keeps the consistent correspondences
    std::vector<cv::Point2f> last_kf_pts, current_kf_pts;
    uint32_t s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (int i = 0; i < 120; i++) {
      const float X = rnd() * 2.f - 1.f, Y = rnd() * 1.4f - 0.7f, Z = 1.5f + rnd();
      last_kf_pts.push_back(cv::Point2f(600.f * X / Z + 320.f, 600.f * Y / Z + 240.f));
      if (i % 6 == 5) current_kf_pts.push_back(cv::Point2f(rnd() * 640.f, rnd() * 480.f));
      else current_kf_pts.push_back(cv::Point2f(600.f * (X - 0.05f) / (Z - 0.02f) + 320.f, 600.f * (Y + 0.01f) / (Z - 0.02f) + 240.f));
    }
    std::vector<uchar> kf_inliers_mask;
    cv::Mat F = dvslam::findFundamentalMat(matcher_, last_kf_pts, current_kf_pts, kf_inliers_mask, dvslam::FM_RANSAC, 2.0, 0.99);
    int kept = 0, kept_outliers = 0;
    for (size_t i = 0; i < kf_inliers_mask.size(); i++) if (kf_inliers_mask[i]) { kept++; kept_outliers += i % 6 == 5; }
    if (F.rows != 3 || kf_inliers_mask.size() != 120 || kept < 85 || kept_outliers > 4) { std::printf("findFundamentalMat: %d kept, %d of them outliers\n", kept, kept_outliers); return 1; }
    std::vector<cv::Point2f> few(last_kf_pts.begin(), last_kf_pts.begin() + 10), few2(current_kf_pts.begin(), current_kf_pts.begin() + 10);
    dvslam::findFundamentalMat(matcher_, few, few2, kf_inliers_mask, dvslam::FM_RANSAC, 2.0, 0.99);   // below 15 points: LMedS, as OpenCV switches
    if (kf_inliers_mask.size() != 10) return 1;
    std::printf("dvslam::findFundamentalMat: %d of 120 kept\n", kept);
  }
  {  // frontend.cpp:905-925 with cv::solvePnPRansac -> dvslam::solvePnPRansac(matcher_, ...): a small known motion, every sixth correspondence wrong
    std::vector<cv::Point3f> points3d; std::vector<cv::Point2f> points2d;
    uint32_t s = 7;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (int i = 0; i < 150; i++) {
      const float X = rnd() * 2.f - 1.f, Y = rnd() * 1.4f - 0.7f, Z = 1.2f + 1.5f * rnd();
      points3d.push_back(cv::Point3f(X, Y, Z));
      if (i % 6 == 5) points2d.push_back(cv::Point2f(rnd() * 640.f, rnd() * 480.f));
      else points2d.push_back(cv::Point2f(600.f * (X + 0.04f) / (Z - 0.03f) + 320.f, 600.f * (Y - 0.02f) / (Z - 0.03f) + 240.f));   // t = (0.04, -0.02, -0.03), R = I
    }
    cv::Mat K(3, 3, CV_64F), dist, rvec, tvec;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K.at<double>(i, j) = 0.0;
    K.at<double>(0, 0) = 600; K.at<double>(1, 1) = 600; K.at<double>(0, 2) = 320; K.at<double>(1, 2) = 240; K.at<double>(2, 2) = 1;
    std::vector<int> inliers;
    const bool success = dvslam::solvePnPRansac(matcher_, points3d, points2d, K, dist, rvec, tvec, false, 100, 4.0, 0.99, &inliers);
    const double et = std::fabs(tvec.at<double>(0, 0) - 0.04) + std::fabs(tvec.at<double>(1, 0) + 0.02) + std::fabs(tvec.at<double>(2, 0) + 0.03);
    const double er = std::fabs(rvec.at<double>(0, 0)) + std::fabs(rvec.at<double>(1, 0)) + std::fabs(rvec.at<double>(2, 0));
    if (!success || inliers.size() < 120 || inliers.size() > 130 || et > 1e-3 || er > 1e-3) { std::printf("solvePnPRansac: %d, %zu inliers, |dt| %.2e |r| %.2e\n", (int)success, inliers.size(), et, er); return 1; }
    std::printf("dvslam::solvePnPRansac: %zu of 150 inliers\n", inliers.size());
    // the same scene through a camera that reports plumb_bob coefficients (camera_info's D, frontend.cpp:911-921), here as CV_32F: the
    // adapter moves the points to distortion-free pixels first and finds the same motion
    const double Dd[5] = {-0.28, 0.07, 0.0012, -0.0008, 0.004};
    cv::Mat distf(1, 5, CV_32F);
    for (int k = 0; k < 5; k++) distf.at<float>(0, k) = (float)Dd[k];
    std::vector<cv::Point2f> distorted;
    for (size_t i = 0; i < points2d.size(); i++) {
      const double x = (points2d[i].x - 320.0) / 600.0, y = (points2d[i].y - 240.0) / 600.0, r2 = x * x + y * y;
      const double cd = 1 + ((Dd[4] * r2 + Dd[1]) * r2 + Dd[0]) * r2;
      const double xd = x * cd + 2 * Dd[2] * x * y + Dd[3] * (r2 + 2 * x * x), yd = y * cd + Dd[2] * (r2 + 2 * y * y) + 2 * Dd[3] * x * y;
      distorted.push_back(cv::Point2f((float)(xd * 600.0 + 320.0), (float)(yd * 600.0 + 240.0)));
    }
    cv::Mat rvec2, tvec2;
    std::vector<int> inliers2;
    const bool success2 = dvslam::solvePnPRansac(matcher_, points3d, distorted, K, distf, rvec2, tvec2, false, 100, 4.0, 0.99, &inliers2);
    const double et2 = std::fabs(tvec2.at<double>(0, 0) - 0.04) + std::fabs(tvec2.at<double>(1, 0) + 0.02) + std::fabs(tvec2.at<double>(2, 0) + 0.03);
    if (!success2 || inliers2.size() < 118 || inliers2.size() > 130 || et2 > 2e-3) { std::printf("solvePnPRansac (distorted): %d, %zu inliers, |dt| %.2e\n", (int)success2, inliers2.size(), et2); return 1; }
    std::printf("dvslam::solvePnPRansac with D != 0: %zu of 150 inliers\n", inliers2.size());
  }
  std::printf("opencv-typed adapters ok: %d keypoints, BA cost %.3e in %d steps\n", n, result.final_cost, result.iterations_completed);
  return 0;
}
