// dvslam/sliding_window_ba.hpp — adapter with the call surface of the reference's SlidingWindowBA
// (include/dynamic_visual_slam/bundle_adjustment.hpp:652-904) over the C-ABI.  It reproduces optimize()'s host
// logic in behaviour: fromRt/toRt pose conversion (:138-212), frame-id / landmark-id maps (landmarks keyed by
// id only, :762/:790), skipping observations with unknown ids (:805-809), the two early-exit messages (:750-755,
// :829-834), first keyframe of the vector as gauge (:781-785), success <=> CONVERGENCE (:860), results keyed
// (id, category) (:881-888).
//
// Two front ends share that logic (detail::optimize_impl):
//   dvslam::SlidingWindowBA            poses as plain row-major double[9] / double[3]: no OpenCV / ROS dependency (this file);
//   ::SlidingWindowBA, ::KeyframeData… the reference's own global-namespace types with cv::Mat / cv::Point3d / rclcpp::Time,
//                                      in include/dynamic_visual_slam/bundle_adjustment.hpp — the same include path as the
//                                      reference header, so backend.cpp:180, 661, 908-973, 1356-1392 compile unchanged.
// One dvs_ba handle (HIP stream, pinned blocks, two grow-only device arenas) lives as long as the adapter object: a window whose
// shape fits the arenas allocates nothing — dvs_ba_set_problem is one pinned staging copy + one fill on the handle's stream
// (0.11 ms for 10 keyframes x 2000 landmarks, tools/time_ba_setup.py).
#pragma once
#include <array>
#include <chrono>
#include <cstdio>
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

namespace detail {

// owns the dvs_ba handle for the adapter's lifetime (created at first use so that constructing an adapter never touches the GPU)
class BaEngine {
 public:
  explicit BaEngine(int device) : device_(device) {}
  ~BaEngine() { dvs_ba_destroy(h_); }
  BaEngine(const BaEngine&) = delete;
  BaEngine& operator=(const BaEngine&) = delete;
  bool warned_host_solver = false;   // the fall-back to host linear algebra announces itself once per object
  int last_linear_solver = 0;        // dvs_ba_summary::linear_solver of the last optimize(): 1 = device, 2 = host, 0 = none ran
  dvs_ba* get() {
    if (!h_ && dvs_ba_create(device_, &h_) != DVS_OK) h_ = nullptr;
    return h_;
  }

 private:
  dvs_ba* h_ = nullptr;
  int device_;
};

struct Intrinsics { double fx, fy, cx, cy, sigma; };

// Traits: Result type + how to read a keyframe's (R, t) and write the optimised poses / landmarks in the front end's types
template <class Traits, class KF, class LM, class OB>
typename Traits::Result optimize_impl(BaEngine& eng, const Intrinsics& K_, const std::vector<KF>& keyframes, const std::vector<LM>& landmarks,
                                      const std::vector<OB>& observations, int max_iterations) {
  const auto t0 = std::chrono::high_resolution_clock::now();
  auto elapsed = [&] { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0); };
  typename Traits::Result res;
  res.success = false;
  res.final_cost = 0; res.iterations_completed = 0;
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
    double R9[9], t3[3], qq[4], tr[3];
    Traits::get_rt(kf, R9, t3);
    dvs_ba_pose_from_rt(R9, t3, qq, tr);
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
  dvs_ba* h = eng.get();
  dvs_ba_summary s{};
  bool ok = h != nullptr &&
            dvs_ba_set_problem(h, K, q.data(), t.data(), L, X.data(), (int)cam.size(), cam.data(), lmi.data(), uv.data(), pose_fixed.data(),
                               lm_fixed.data(), K_.fx, K_.fy, K_.cx, K_.cy, K_.sigma, 1.345) == DVS_OK;
  if (ok) {  // linear algebra on the device for sliding-window shapes, GPU evaluation + host Schur complement otherwise
    dvs_status st = dvs_ba_solve_device(h, max_iterations, 1e-6, 1e-10, 1e-8, &s);
    if (st == DVS_ERR_UNSUPPORTED) {
      // NOT silent: the device solver takes windows of up to 16 free keyframes (the reference's live window is 5-10,
      // backend.cpp:895); a larger one still optimises — GPU evaluation, normal equations on the host, ~40x slower per solve — and says so
      // once per object on stderr, in last_linear_solver() and in dvs_ba_summary::linear_solver
      if (!eng.warned_host_solver) {
        std::fprintf(stderr, "dvslam::SlidingWindowBA: %s — solving the normal equations on the host for this window\n", dvs_last_error());
        eng.warned_host_solver = true;
      }
      st = dvs_ba_solve(h, max_iterations, 1e-6, 1e-10, 1e-8, &s);
    }
    ok = st == DVS_OK;
  }
  eng.last_linear_solver = ok ? s.linear_solver : 0;
  if (!ok) {  // the reference never throws: exceptions become success=false + message (:890-895)
    res.message = std::string("Bundle adjustment exception: ") + dvs_last_error();
    res.optimization_time = elapsed();
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
    double R9[9], t3[3];
    dvs_ba_pose_to_rt(&q[4 * fs.second], &t[3 * fs.second], R9, t3);
    Traits::put_pose(res, fs.first, R9, t3);
  }
  for (const auto& ls : lm_slot) Traits::put_landmark(res, ls.first, lm_cat[ls.second], &X[3 * ls.second]);
  res.optimization_time = elapsed();
  return res;
}

struct PlainTraits {
  typedef OptimizationResult Result;
  static void get_rt(const KeyframeData& kf, double* R9, double* t3) { std::memcpy(R9, kf.R, 72); std::memcpy(t3, kf.t, 24); }
  static void put_pose(Result& r, int fid, const double* R9, const double* t3) {
    Pose p;
    std::memcpy(p.R, R9, 72); std::memcpy(p.t, t3, 24);
    r.optimized_poses[fid] = p;
  }
  static void put_landmark(Result& r, uint64_t id, const std::string& cat, const double* X) { r.optimized_landmarks[{id, cat}] = {X[0], X[1], X[2]}; }
};

}  // namespace detail

class SlidingWindowBA {
 public:
  // intrinsics and sigma exactly as passed: the backend really calls SlidingWindowBA(10, fx, fy, cx, cy) (backend.cpp:180,661)
  SlidingWindowBA(double fx, double fy, double cx, double cy, double sigma_pixels = 1.0, int device = 0)
      : k_{fx, fy, cx, cy, sigma_pixels}, eng_(device) {}

  OptimizationResult optimize(const std::vector<KeyframeData>& keyframes, const std::vector<Landmark>& landmarks,
                              const std::vector<Observation>& observations, int max_iterations = 10) {
    return detail::optimize_impl<detail::PlainTraits>(eng_, k_, keyframes, landmarks, observations, max_iterations);
  }
  // (not in the reference) who solved the normal equations of the last optimize(): 1 = the device, 2 = the host (> 16 free keyframes), 0 = nobody
  int last_linear_solver() const { return eng_.last_linear_solver; }

 private:
  detail::Intrinsics k_;
  detail::BaEngine eng_;
};

}  // namespace dvslam
