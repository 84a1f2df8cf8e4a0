// dynamic_visual_slam/bundle_adjustment.hpp — source-compatible replacement for the reference header of the same include
// path (reference include/dynamic_visual_slam/bundle_adjustment.hpp:92-432, 652-904): the global-namespace types the backend
// compiles against — CameraPose, Landmark, Observation, KeyframeData(int, cv::Mat, cv::Mat, rclcpp::Time),
// OptimizationResult with std::pair<cv::Mat, cv::Mat> poses and cv::Point3d landmarks, SlidingWindowBA — with
// optimize() running on the MI355X through the C-ABI (dvs_ba_*) instead of Ceres.  Putting this repo's include/ in
// front of the reference's on the include path is the whole backend-side integration: backend.cpp:180, 661 (ctor),
// 908-960 (window assembly + optimize), 967-977 (result fields), 1356-1392 (updateOptimizedResults) compile unchanged.
// Needs the headers the reference's own file needs minus Ceres / Eigen: OpenCV core and rclcpp (for rclcpp::Time).
#ifndef BUNDLE_ADJUSTMENT_HPP
#define BUNDLE_ADJUSTMENT_HPP

#include <opencv2/core/core.hpp>
#include <chrono>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "rclcpp/rclcpp.hpp"
#include "../dvslam/sliding_window_ba.hpp"

// optimiser-side pose: world->camera quaternion (w, x, y, z) + translation (reference :92-213)
struct CameraPose {
  double rotation[4];
  double translation[3];
  CameraPose() : rotation{1.0, 0.0, 0.0, 0.0}, translation{0.0, 0.0, 0.0} {}
  void fromRt(const cv::Mat& R_world_camera, const cv::Mat& t_world_camera) {
    double R9[9], t3[3];
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R9[3 * i + j] = R_world_camera.at<double>(i, j);
      t3[i] = t_world_camera.at<double>(i);
    }
    dvs_ba_pose_from_rt(R9, t3, rotation, translation);
  }
  void toRt(cv::Mat& R_world_camera, cv::Mat& t_world_camera) const {
    double R9[9], t3[3];
    dvs_ba_pose_to_rt(rotation, translation, R9, t3);
    R_world_camera = cv::Mat(3, 3, CV_64F);
    t_world_camera = cv::Mat(3, 1, CV_64F);
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R_world_camera.at<double>(i, j) = R9[3 * i + j];
      t_world_camera.at<double>(i) = t3[i];
    }
  }
};

using Landmark = dvslam::Landmark;        // same fields and constructors as the reference's (:238-282)
using Observation = dvslam::Observation;  // (:308-338)

struct KeyframeData {  // (:362-389)
  int frame_id;
  cv::Mat R;
  cv::Mat t;
  rclcpp::Time timestamp;
  KeyframeData(int id, const cv::Mat& rotation, const cv::Mat& translation, const rclcpp::Time& stamp)
      : frame_id(id), R(rotation.clone()), t(translation.clone()), timestamp(stamp) {}
};

struct OptimizationResult {  // (:419-432)
  bool success;
  double final_cost;
  int iterations_completed;
  int frames_optimized;
  int landmarks_optimized;
  std::string message;
  std::chrono::milliseconds optimization_time;
  std::map<int, std::pair<cv::Mat, cv::Mat>> optimized_poses;
  std::map<std::pair<uint64_t, std::string>, cv::Point3d> optimized_landmarks;
};

class SlidingWindowBA {  // (:652-904)
 public:
  SlidingWindowBA(double fx, double fy, double cx, double cy, double sigma_pixels = 1.0) : k_{fx, fy, cx, cy, sigma_pixels}, eng_(0) {}

  OptimizationResult optimize(const std::vector<KeyframeData>& keyframes, const std::vector<Landmark>& landmarks,
                              const std::vector<Observation>& observations, int max_iterations = 10) {
    return dvslam::detail::optimize_impl<CvTraits>(eng_, k_, keyframes, landmarks, observations, max_iterations);
  }
  // (not in the reference) who solved the normal equations of the last optimize(): 1 = the device, 2 = the host (> 16 free keyframes), 0 = nobody
  int last_linear_solver() const { return eng_.last_linear_solver; }

 private:
  struct CvTraits {
    typedef OptimizationResult Result;
    static void get_rt(const KeyframeData& kf, double* R9, double* t3) {
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R9[3 * i + j] = kf.R.at<double>(i, j);
        t3[i] = kf.t.at<double>(i);
      }
    }
    static void put_pose(Result& r, int fid, const double* R9, const double* t3) {
      cv::Mat R(3, 3, CV_64F), t(3, 1, CV_64F);
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R.at<double>(i, j) = R9[3 * i + j];
        t.at<double>(i) = t3[i];
      }
      r.optimized_poses[fid] = std::make_pair(R, t);
    }
    static void put_landmark(Result& r, uint64_t id, const std::string& cat, const double* X) {
      r.optimized_landmarks[std::make_pair(id, cat)] = cv::Point3d(X[0], X[1], X[2]);
    }
  };
  dvslam::detail::Intrinsics k_;
  dvslam::detail::BaEngine eng_;
};

#endif  // BUNDLE_ADJUSTMENT_HPP
