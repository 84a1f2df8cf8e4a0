// dvslam/calib3d.hpp — adapter with the call surface of
//   cv::findFundamentalMat(points1, points2, mask, cv::FM_RANSAC, 2.0, 0.99)        src/frontend.cpp:635, 1146-1147
// over the C-ABI: OpenCV's own procedure (dvs_find_fundamental_cv — the cv::RNG sample sequence, 7-point solver, float error compare;
// RANSAC with the adaptive stop from 15 correspondences on, LMedS below, as cv::findFundamentalMat switches).  Below 8 the reference
// never calls (frontend.cpp:627 `if (... size() >= 8)`): an all-zero mask and an empty F.
//   dvslam::findFundamentalMat(matcher, pts1, pts2, n, mask, F9)                       plain pointers (n x 2 floats)
//   (DVSLAM_WITH_OPENCV) findFundamentalMat(matcher, vector<cv::Point2f>, vector<cv::Point2f>, vector<uchar>& mask, method, ...)
// and of
//   cv::solvePnPRansac(points3d, points2d, K, dist, rvec, tvec, false, 100, 4.0, 0.99, inliers)                        src/frontend.cpp:911-921
// (dvs_solve_pnp_ransac_cv — OpenCV's procedure with its default flags, restated from the published sources, PARITY UNPINNED: cv::RNG
// 5-point samples, EPnP, float scoring, adaptive stop, solvePnP(ITERATIVE) on the inliers).  The C-ABI entry takes a pinhole camera.  The
// reference passes rgb_dist_coeffs_ from camera_info's D (frontend.cpp:911-921): NON-ZERO coefficients are handled HERE, on the host, by
// moving the image points to the distortion-free pixels first (undistortImagePoints below: cv::undistortPoints' iteration with P = K;
// n <= 2000 points, microseconds) — the threshold then applies in rectified pixels, where OpenCV applies it to the distorted projection.
// Fewer than 6 correspondences: false (the reference returns before the call, frontend.cpp:900).
//   dvslam::solvePnPRansac(matcher, obj, img, n, K4, rvec, tvec, inliers)                plain pointers (n x 3 / n x 2 floats)
//   (DVSLAM_WITH_OPENCV) solvePnPRansac(matcher, vector<cv::Point3f>, vector<cv::Point2f>, cv::Mat K, cv::Mat dist, cv::Mat& rvec, ...)
// `matcher` is the dvslam::BFMatcher the node already owns (frontend.cpp:220): the stage shares its handle, stream and scratch.
#pragma once
#include <stdexcept>
#include <vector>
#include "bf_matcher.hpp"

namespace dvslam {

constexpr int FM_RANSAC = 8;   // cv::FM_RANSAC

// returns true when a model was found; F9 row-major (may be nullptr)
inline bool findFundamentalMat(BFMatcher& matcher, const float* pts1, const float* pts2, int n, std::vector<unsigned char>& mask, double* F9 = nullptr,
                               double ransacReprojThreshold = 2.0, double confidence = 0.99, int maxIters = 1000) {
  mask.assign((size_t)(n > 0 ? n : 0), 0);
  double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int32_t inliers = 0;
  if (n >= 8 &&
      dvs_find_fundamental_cv(matcher.handle(), pts1, pts2, n, ransacReprojThreshold, confidence, maxIters, F, mask.data(), &inliers, nullptr) != DVS_OK)
    throw std::runtime_error(dvs_last_error());
  if (F9) for (int k = 0; k < 9; k++) F9[k] = F[k];
  bool any = false;
  for (int k = 0; k < 9; k++) any = any || F[k] != 0.0;
  return any;   // false: OpenCV would return an empty matrix (the mask is written either way)
}

// returns OpenCV's return value; rvec / tvec as it leaves them (the RANSAC stage's model when only the refit failed); inliers in index order
inline bool solvePnPRansac(BFMatcher& matcher, const float* objectPoints, const float* imagePoints, int n, const double K4[4] /* fx fy cx cy */, double rvec[3],
                           double tvec[3], std::vector<int>& inliers, int iterationsCount = 100, float reprojectionError = 8.0f, double confidence = 0.99) {
  inliers.assign((size_t)(n > 0 ? n : 0), 0);
  int32_t nin = 0, ok = 0;
  if (n > 0 && dvs_solve_pnp_ransac_cv(matcher.handle(), objectPoints, imagePoints, n, K4, iterationsCount, reprojectionError, confidence, rvec, tvec,
                                       inliers.data(), &nin, &ok, nullptr) != DVS_OK)
    throw std::runtime_error(dvs_last_error());
  inliers.resize((size_t)nin);
  return ok != 0;
}

// Image points -> the pixels a distortion-free camera with the same K would see (cv::undistortPoints(src, dst, K, D, noArray(), K) as
// OpenCV 4.x runs it: x0 = (u - cx) / fx, five fixed-point passes  x <- (x0 - dx(x)) / (1 + k1 r2 + k2 r4 + k3 r6) * (1 + k4 r2 + k5 r4 + k6 r6),
// dx = 2 p1 x y + p2 (r2 + 2 x x) [dy alike], back through K).  D = (k1, k2, p1, p2[, k3[, k4, k5, k6]]): 4, 5 or 8 coefficients — what
// sensor_msgs/CameraInfo's plumb_bob and rational_polynomial models carry; thin-prism / tilt terms (12, 14) are refused.
inline void undistortImagePoints(const float* imagePoints, int n, const double K4[4], const double* D, int nD, std::vector<float>& out) {
  if (!(nD == 0 || nD == 4 || nD == 5 || nD == 8)) throw std::invalid_argument("dvslam::undistortImagePoints: 4, 5 or 8 distortion coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6]])");
  out.assign(imagePoints, imagePoints + 2 * (size_t)(n > 0 ? n : 0));
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool any = false;
  for (int i = 0; i < nD; i++) { k[i] = D[i]; any = any || D[i] != 0.0; }
  if (!any) return;
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  for (int i = 0; i < n; i++) {
    const double x0 = ((double)imagePoints[2 * i] - cx) / fx, y0 = ((double)imagePoints[2 * i + 1] - cy) / fy;
    double x = x0, y = y0;
    for (int it = 0; it < 5; it++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x), dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - dx) * icdist; y = (y0 - dy) * icdist;
    }
    out[2 * (size_t)i] = (float)(x * fx + cx); out[2 * (size_t)i + 1] = (float)(y * fy + cy);
  }
}

// plain pointers with distortion: D as above (nullptr / 0: none)
inline bool solvePnPRansac(BFMatcher& matcher, const float* objectPoints, const float* imagePoints, int n, const double K4[4], const double* D, int nD, double rvec[3],
                           double tvec[3], std::vector<int>& inliers, int iterationsCount = 100, float reprojectionError = 8.0f, double confidence = 0.99) {
  std::vector<float> und;
  undistortImagePoints(imagePoints, n, K4, D, nD, und);
  return solvePnPRansac(matcher, objectPoints, und.data(), n, K4, rvec, tvec, inliers, iterationsCount, reprojectionError, confidence);
}

#ifdef DVSLAM_WITH_OPENCV
// the reference's call with the matcher in front; K: 3 x 3 CV_64F or CV_32F, distCoeffs: empty or 4 / 5 / 8 coefficients of either type
// (camera_info's D, frontend.cpp:911-921); rvec / tvec: 3 x 1 CV_64F
inline bool solvePnPRansac(BFMatcher& matcher, const std::vector<cv::Point3f>& objectPoints, const std::vector<cv::Point2f>& imagePoints, const cv::Mat& cameraMatrix,
                           const cv::Mat& distCoeffs, cv::Mat& rvec, cv::Mat& tvec, bool useExtrinsicGuess = false, int iterationsCount = 100,
                           float reprojectionError = 8.0f, double confidence = 0.99, std::vector<int>* inliers = nullptr) {
  CV_Assert(!useExtrinsicGuess && objectPoints.size() == imagePoints.size() && cameraMatrix.rows == 3 && cameraMatrix.cols == 3 &&
            (cameraMatrix.type() == CV_64F || cameraMatrix.type() == CV_32F));
  CV_Assert(distCoeffs.empty() || distCoeffs.type() == CV_64F || distCoeffs.type() == CV_32F);
  static_assert(sizeof(cv::Point3f) == 3 * sizeof(float), "cv::Point3f is three packed floats");
  auto Kat = [&](int i, int j) { return cameraMatrix.type() == CV_64F ? cameraMatrix.at<double>(i, j) : (double)cameraMatrix.at<float>(i, j); };
  const double K4[4] = {Kat(0, 0), Kat(1, 1), Kat(0, 2), Kat(1, 2)};
  std::vector<double> D(distCoeffs.empty() ? 0 : distCoeffs.total());
  for (size_t i = 0; i < D.size(); i++) D[i] = distCoeffs.type() == CV_64F ? distCoeffs.at<double>((int)i) : (double)distCoeffs.at<float>((int)i);
  double r[3] = {0, 0, 0}, t[3] = {0, 0, 0};
  std::vector<int> inl;
  const bool ok = solvePnPRansac(matcher, objectPoints.empty() ? nullptr : &objectPoints[0].x, imagePoints.empty() ? nullptr : &imagePoints[0].x,
                                 (int)objectPoints.size(), K4, D.empty() ? nullptr : D.data(), (int)D.size(), r, t, inl, iterationsCount, reprojectionError,
                                 confidence);
  rvec = cv::Mat(3, 1, CV_64F); tvec = cv::Mat(3, 1, CV_64F);
  for (int k = 0; k < 3; k++) { rvec.at<double>(k, 0) = r[k]; tvec.at<double>(k, 0) = t[k]; }
  if (inliers) *inliers = ok ? inl : std::vector<int>();      // OpenCV releases the inlier array when it returns false
  return ok;
}

// the reference's call, with the matcher in front: `cv::findFundamentalMat(a, b, mask, cv::FM_RANSAC, 2.0, 0.99)` becomes
// `dvslam::findFundamentalMat(matcher_, a, b, mask, cv::FM_RANSAC, 2.0, 0.99)`; returns F as a 3 x 3 CV_64F matrix (empty: none)
inline cv::Mat findFundamentalMat(BFMatcher& matcher, const std::vector<cv::Point2f>& points1, const std::vector<cv::Point2f>& points2,
                                  std::vector<unsigned char>& mask, int method = FM_RANSAC, double ransacReprojThreshold = 2.0, double confidence = 0.99,
                                  int maxIters = 1000) {
  CV_Assert(method == FM_RANSAC && points1.size() == points2.size());   // the only configuration the reference uses
  static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f is two packed floats");
  double F[9];
  const bool ok = findFundamentalMat(matcher, points1.empty() ? nullptr : &points1[0].x, points2.empty() ? nullptr : &points2[0].x, (int)points1.size(), mask, F,
                                     ransacReprojThreshold, confidence, maxIters);
  if (!ok) return cv::Mat();
  cv::Mat out(3, 3, CV_64F);
  for (int k = 0; k < 9; k++) out.at<double>(k / 3, k % 3) = F[k];
  return out;
}
#endif

}  // namespace dvslam
