// dvslam/sliding_window_ba.hpp — adapter with the call surface of the reference's SlidingWindowBA
// (include/dynamic_visual_slam/bundle_adjustment.hpp:652-904) over the C-ABI.  It reproduces optimize()'s host
// logic verbatim in behaviour: fromRt/toRt pose conversion (:138-212), frame-id / landmark-id maps (landmarks keyed by
// id only, :762/:790), skipping observations with unknown ids (:805-809), the two early-exit messages (:750-755,
// :829-834), first keyframe of the vector as gauge (:781-785), success <=> CONVERGENCE (:860), results keyed
// (id, category) (:881-888).  Poses are plain row-major double[9] / double[3] instead of cv::Mat so the header has no
// OpenCV / ROS dependency; KeyframeData::timestamp (rclcpp::Time, unused by optimize) is a double here.
#pragma once
#include <array>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "../dvslam_hip.h"

namespace dvslam {

struct Landmark {
  uint64_t id = 0;
  std::string category;
  double position[3] = {0, 0, 0};
  bool fixed = false;
  Landmark() = default;
  Landmark(uint64_t landmark_id, std::string cat, double x, double y, double z, bool fixed_point = false)
      : id(landmark_id), category(std::move(cat)), fixed(fixed_point) { position[0] = x; position[1] = y; position[2] = z; }
};
struct Observation {
  double pixel[2];
  uint64_t landmark_id;
  std::string category;
  int frame_id;
  Observation(double x, double y, uint64_t landmark, std::string cat, int frame)
      : landmark_id(landmark), category(std::move(cat)), frame_id(frame) { pixel[0] = x; pixel[1] = y; }
};
struct KeyframeData {
  int frame_id;
  double R[9];  // row-major 3x3, caller's convention (the backend passes camera-to-world, backend.cpp:910)
  double t[3];
  double timestamp;
  KeyframeData(int id, const double* rotation, const double* translation, double stamp = 0.0) : frame_id(id), timestamp(stamp) {
    std::memcpy(R, rotation, sizeof(R)); std::memcpy(t, translation, sizeof(t));
  }
};
struct Pose { double R[9]; double t[3]; };
struct OptimizationResult {
  bool success = false;
  double final_cost = 0;
  int iterations_completed = 0, frames_optimized = 0, landmarks_optimized = 0;
  std::string message;
  std::chrono::milliseconds optimization_time{0};
  std::map<int, Pose> optimized_poses;
  std::map<std::pair<uint64_t, std::string>, std::array<double, 3>> optimized_landmarks;
};

class SlidingWindowBA {
 public:
  // intrinsics and sigma exactly as passed: the backend really calls SlidingWindowBA(10, fx, fy, cx, cy) (backend.cpp:180,661)
  SlidingWindowBA(double fx, double fy, double cx, double cy, double sigma_pixels = 1.0, int device = 0)
      : fx_(fx), fy_(fy), cx_(cx), cy_(cy), sigma_(sigma_pixels), device_(device) {}

  OptimizationResult optimize(const std::vector<KeyframeData>& keyframes, const std::vector<Landmark>& landmarks,
                              const std::vector<Observation>& observations, int max_iterations = 10) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    auto elapsed = [&] { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0); };
    OptimizationResult res;
    res.frames_optimized = (int)keyframes.size();
    res.landmarks_optimized = (int)landmarks.size();
    if (keyframes.empty() || landmarks.empty() || observations.empty()) {
      res.message = "Insufficient input data for optimization";
      res.optimization_time = elapsed();
      return res;
    }
    std::map<int, int> frame_slot;
    std::vector<double> q, t;
    for (const auto& kf : keyframes) {
      double qq[4], tr[3];
      dvs_ba_pose_from_rt(kf.R, kf.t, qq, tr);
      auto it = frame_slot.find(kf.frame_id);
      if (it == frame_slot.end()) {
        frame_slot[kf.frame_id] = (int)q.size() / 4;
        q.insert(q.end(), qq, qq + 4); t.insert(t.end(), tr, tr + 3);
      } else {  // duplicate frame id: the later keyframe replaces the map entry (:769)
        std::memcpy(&q[4 * it->second], qq, sizeof(qq)); std::memcpy(&t[3 * it->second], tr, sizeof(tr));
      }
    }
    std::map<uint64_t, int> lm_slot;
    std::vector<double> X;
    std::vector<uint8_t> lm_fixed;
    std::vector<std::string> lm_cat;
    for (const auto& lm : landmarks) {
      auto it = lm_slot.find(lm.id);
      if (it == lm_slot.end()) {
        lm_slot[lm.id] = (int)lm_fixed.size();
        X.insert(X.end(), lm.position, lm.position + 3); lm_fixed.push_back(lm.fixed ? 1 : 0); lm_cat.push_back(lm.category);
      } else {  // same id in two categories aliases (:790)
        std::memcpy(&X[3 * it->second], lm.position, 3 * sizeof(double)); lm_fixed[it->second] = lm.fixed ? 1 : 0; lm_cat[it->second] = lm.category;
      }
    }
    std::vector<int32_t> cam, lmi;
    std::vector<double> uv;
    for (const auto& ob : observations) {
      auto f = frame_slot.find(ob.frame_id);
      auto l = lm_slot.find(ob.landmark_id);
      if (f != frame_slot.end() && l != lm_slot.end()) { cam.push_back(f->second); lmi.push_back(l->second); uv.push_back(ob.pixel[0]); uv.push_back(ob.pixel[1]); }
    }
    if (cam.empty()) {
      res.message = "No valid observation constraints";
      res.optimization_time = elapsed();
      return res;
    }
    const int K = (int)q.size() / 4, L = (int)lm_fixed.size();
    std::vector<uint8_t> pose_fixed(K, 0);
    pose_fixed[frame_slot[keyframes[0].frame_id]] = 1;
    dvs_ba* h = nullptr;
    dvs_ba_summary s{};
    bool ok = dvs_ba_create(device_, &h) == DVS_OK &&
              dvs_ba_set_problem(h, K, q.data(), t.data(), L, X.data(), (int)cam.size(), cam.data(), lmi.data(), uv.data(), pose_fixed.data(),
                                 lm_fixed.data(), fx_, fy_, cx_, cy_, sigma_, 1.345) == DVS_OK;
    if (ok) {  // linear algebra on the device for sliding-window shapes, GPU evaluation + host Schur complement otherwise
      dvs_status st = dvs_ba_solve_device(h, max_iterations, 1e-6, 1e-10, 1e-8, &s);
      if (st == DVS_ERR_UNSUPPORTED) st = dvs_ba_solve(h, max_iterations, 1e-6, 1e-10, 1e-8, &s);
      ok = st == DVS_OK;
    }
    if (!ok) {  // the reference never throws: exceptions become success=false + message (:890-895)
      res.message = std::string("Bundle adjustment exception: ") + dvs_last_error();
      res.optimization_time = elapsed();
      dvs_ba_destroy(h);
      return res;
    }
    static const char* kTerm[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    res.success = s.termination == 0;
    res.final_cost = s.final_cost;
    res.iterations_completed = s.num_successful_steps;
    res.message = res.success ? "Bundle adjustment converged successfully"
                              : std::string("Bundle adjustment failed to converge: ") + kTerm[s.termination < 0 || s.termination > 2 ? 2 : s.termination];
    dvs_ba_get_parameters(h, q.data(), t.data(), X.data());
    for (const auto& fs : frame_slot) {
      Pose p;
      dvs_ba_pose_to_rt(&q[4 * fs.second], &t[3 * fs.second], p.R, p.t);
      res.optimized_poses[fs.first] = p;
    }
    for (const auto& ls : lm_slot) res.optimized_landmarks[{ls.first, lm_cat[ls.second]}] = {X[3 * ls.second], X[3 * ls.second + 1], X[3 * ls.second + 2]};
    dvs_ba_destroy(h);
    res.optimization_time = elapsed();
    return res;
  }

 private:
  double fx_, fy_, cx_, cy_, sigma_;
  int device_;
};

}  // namespace dvslam
