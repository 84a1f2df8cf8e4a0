// TEST-ONLY stand-in for the one rclcpp type the BA adapter stores (KeyframeData::timestamp, reference
// bundle_adjustment.hpp:366); see ../opencv2/core/core.hpp for why these exist.
#pragma once
#include <cstdint>
namespace rclcpp {
class Time {
 public:
  Time() {}
  explicit Time(int64_t ns) : ns_(ns) {}
  int64_t nanoseconds() const { return ns_; }
 private:
  int64_t ns_ = 0;
};
}  // namespace rclcpp
