// dvslam/orb_extractor.hpp — header-only C++ adapter with the call surface of the reference's
// ORB_SLAM3::ORBextractor (include/dynamic_visual_slam/ORBextractor.hpp:44-110) over the C-ABI in
// dvslam_hip.h.  Two layers:
//   dvslam::OrbExtractor      plain pointers, std::vector<dvs_keypoint>; needs nothing but the C-ABI
//   ORB_SLAM3::ORBextractor   the reference's own signature (cv::InputArray, std::vector<cv::KeyPoint>,
//                             cv::OutputArray); compiled only when DVSLAM_WITH_OPENCV is defined, so the
//                             frontend (src/frontend.cpp:205-211, 1094-1095, 1285-1286) builds unchanged.
// Error behaviour mirrors the reference: operator() returns -1 for an empty image
// (ORBextractor.cpp:1090-1091), otherwise the number of keypoints; other failures throw.
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../dvslam_hip.h"
#ifdef DVSLAM_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace dvslam {

class OrbExtractor {
 public:
  OrbExtractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0, int max_batch = 1)
      : nlevels_(nlevels), scale_factor_(scaleFactor) {
    dvs_orb_params p{};
    p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels;
    p.ini_th_fast = iniThFAST; p.min_th_fast = minThFAST; p.max_batch = max_batch;
    if (dvs_orb_create(&p, device, &h_) != DVS_OK) throw std::runtime_error(std::string("dvs_orb_create: ") + dvs_last_error());
    cap_ = dvs_orb_max_keypoints(h_);
  }
  ~OrbExtractor() { dvs_orb_destroy(h_); }
  OrbExtractor(const OrbExtractor&) = delete;
  OrbExtractor& operator=(const OrbExtractor&) = delete;

  // returns the number of keypoints, or -1 for an empty image; descriptors is resized to n*32 bytes
  int operator()(const uint8_t* gray, int rows, int cols, size_t step, std::vector<dvs_keypoint>& keypoints,
                 std::vector<uint8_t>& descriptors) {
    keypoints.assign(cap_, dvs_keypoint{});
    descriptors.assign((size_t)cap_ * 32, 0);
    int32_t n = 0;
    const dvs_status st = dvs_orb_extract(h_, gray, rows, cols, step, keypoints.data(), descriptors.data(), cap_, &n);
    if (st == DVS_ERR_EMPTY) { keypoints.clear(); descriptors.clear(); return -1; }
    if (st != DVS_OK) throw std::runtime_error(std::string("dvs_orb_extract: ") + dvs_last_error());
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    return n;
  }

  int GetLevels() const { return nlevels_; }
  float GetScaleFactor() const { return scale_factor_; }
  std::vector<float> GetScaleFactors() const { return table(0); }
  std::vector<float> GetInverseScaleFactors() const { return table(1); }
  std::vector<float> GetScaleSigmaSquares() const { return table(2); }
  std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
  // mvImagePyramid[level] of the last call (public member in the reference, ORBextractor.hpp:84)
  std::vector<uint8_t> PyramidLevel(int rows, int cols, int level, int* lrows = nullptr, int* lcols = nullptr) const {
    int32_t r = 0, c = 0;
    dvs_orb_level_size(h_, rows, cols, level, &r, &c);
    std::vector<uint8_t> img((size_t)r * c);
    if (dvs_orb_get_level(h_, 0, level, 0, img.data(), (int32_t)img.size()) != DVS_OK) throw std::runtime_error(dvs_last_error());
    if (lrows) *lrows = r;
    if (lcols) *lcols = c;
    return img;
  }
  dvs_orb* handle() { return h_; }
  int capacity() const { return cap_; }

 private:
  std::vector<float> table(int which) const {
    std::vector<float> t[4];
    for (auto& v : t) v.resize(nlevels_);
    dvs_orb_get_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr, nullptr);
    return t[which];
  }
  dvs_orb* h_ = nullptr;
  int nlevels_, cap_ = 0;
  float scale_factor_;
};

}  // namespace dvslam

#ifdef DVSLAM_WITH_OPENCV
namespace ORB_SLAM3 {
// Same name, constructor, operator(), getters and public mvImagePyramid member as the reference class
// (ORBextractor.hpp:44-110), so including this instead of the reference's header is the whole integration on the frontend
// side (include/dynamic_visual_slam/ORBextractor.hpp of this repo does exactly that under the reference's include path).
class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : impl_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) {}
  ~ORBextractor() {}
  int operator()(cv::InputArray _image, cv::InputArray /*_mask: ignored, ORBextractor.hpp:57*/, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& /*vLappingArea = {0,0}: stereo branch never fires*/) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    std::vector<dvs_keypoint> kps;
    std::vector<uint8_t> desc;
    const int n = impl_(image.data, image.rows, image.cols, image.step, kps, desc);
    if (keep_pyramid_) {  // the reference leaves the pyramid of the last frame in this public member (ORBextractor.cpp:1169-1194)
      mvImagePyramid.resize(impl_.GetLevels());
      for (int l = 0; l < impl_.GetLevels(); l++) {
        int lr = 0, lc = 0;
        const std::vector<uint8_t> px = impl_.PyramidLevel(image.rows, image.cols, l, &lr, &lc);
        mvImagePyramid[l].create(lr, lc, CV_8UC1);
        std::memcpy(mvImagePyramid[l].data, px.data(), px.size());
      }
    }
    if (n <= 0) { _descriptors.release(); _keypoints.clear(); return n; }
    _descriptors.create(n, 32, CV_8U);
    std::memcpy(_descriptors.getMat().data, desc.data(), desc.size());
    _keypoints.resize(n);
    for (int i = 0; i < n; i++)
      _keypoints[i] = cv::KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id);
    return n;
  }
  int GetLevels() { return impl_.GetLevels(); }
  float GetScaleFactor() { return impl_.GetScaleFactor(); }
  std::vector<float> GetScaleFactors() { return impl_.GetScaleFactors(); }
  std::vector<float> GetInverseScaleFactors() { return impl_.GetInverseScaleFactors(); }
  std::vector<float> GetScaleSigmaSquares() { return impl_.GetScaleSigmaSquares(); }
  std::vector<float> GetInverseScaleSigmaSquares() { return impl_.GetInverseScaleSigmaSquares(); }

  // level images of the last frame (continuous 8UC1; the reference's are views into a bordered buffer, same pixels).  The
  // reference's frontend never reads it, and filling it costs one device-to-host copy of the pyramid (2.85 MB at 720p) per
  // frame: callers that do not need it switch the copy off.
  std::vector<cv::Mat> mvImagePyramid;
  void keepImagePyramid(bool on) { keep_pyramid_ = on; if (!on) mvImagePyramid.clear(); }

 private:
  dvslam::OrbExtractor impl_;
  bool keep_pyramid_ = true;
};
}  // namespace ORB_SLAM3
#endif
