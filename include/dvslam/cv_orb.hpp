// dvslam/cv_orb.hpp — header-only C++ adapter with the call surface of cv::ORB as the reference uses it
// (/root/reference/dynamic_visual_slam/test/test_dbow2_integration.cpp:19 `cv::ORB::create(100)`, :38
// `orb_->detectAndCompute(dummy_image_, cv::Mat(), keypoints, descriptors)`) over dvs_cvorb_* of dvslam_hip.h.  Two layers:
//   dvslam::CvOrb   plain pointers, std::vector<dvs_keypoint>; needs nothing but the C-ABI
//   dvslam::ORB     cv::ORB's create() / detectAndCompute() signatures (cv::Ptr, cv::InputArray, std::vector<cv::KeyPoint>,
//                   cv::OutputArray); compiled only when DVSLAM_WITH_OPENCV is defined.  A call site changes `cv::ORB` to
//                   `dvslam::ORB`; nothing else.
// Behaviour mirrors cv::ORB: an empty image leaves keypoints / descriptors untouched; no keypoints -> descriptors released;
// a non-empty mask and useProvidedKeypoints are not built (they throw).
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../dvslam_hip.h"
#ifdef DVSLAM_WITH_OPENCV
#include <memory>
#include <opencv2/core/core.hpp>
#endif

namespace dvslam {

class CvOrb {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  explicit CvOrb(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31, int firstLevel = 0, int WTA_K = 2,
                 int scoreType = HARRIS_SCORE, int patchSize = 31, int fastThreshold = 20, int device = 0) {
    dvs_cvorb_params p{};
    p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels; p.edge_threshold = edgeThreshold; p.first_level = firstLevel;
    p.wta_k = WTA_K; p.score_type = scoreType; p.patch_size = patchSize; p.fast_threshold = fastThreshold;
    if (dvs_cvorb_create(&p, device, &h_) != DVS_OK) throw std::runtime_error(std::string("dvs_cvorb_create: ") + dvs_last_error());
    cap_ = nfeatures + 64;
  }
  ~CvOrb() { dvs_cvorb_destroy(h_); }
  CvOrb(const CvOrb&) = delete;
  CvOrb& operator=(const CvOrb&) = delete;

  // returns the number of keypoints (= rows of the descriptor matrix).  retainBest keeps every keypoint that ties with the last one,
  // so the count can exceed nfeatures: the call is repeated once with the capacity the library asked for.
  int detectAndCompute(const uint8_t* gray, int rows, int cols, size_t step, std::vector<dvs_keypoint>& keypoints, std::vector<uint8_t>& descriptors) {
    for (int attempt = 0; attempt < 2; attempt++) {
      keypoints.assign(cap_, dvs_keypoint{});
      descriptors.assign((size_t)cap_ * 32, 0);
      int32_t n = 0;
      const dvs_status st = dvs_cvorb_detect_and_compute(h_, gray, rows, cols, step, keypoints.data(), descriptors.data(), cap_, &n);
      if (st == DVS_ERR_CAPACITY && attempt == 0) { cap_ = n + 64; continue; }
      if (st != DVS_OK) throw std::runtime_error(std::string("dvs_cvorb_detect_and_compute: ") + dvs_last_error());
      keypoints.resize(n);
      descriptors.resize((size_t)n * 32);
      return n;
    }
    return 0;
  }
  dvs_cvorb* handle() { return h_; }

 private:
  dvs_cvorb* h_ = nullptr;
  int cap_ = 0;
};

#ifdef DVSLAM_WITH_OPENCV
class ORB {
 public:
  enum ScoreType { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  static std::shared_ptr<ORB> create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31, int firstLevel = 0,
                                     int WTA_K = 2, int scoreType = HARRIS_SCORE, int patchSize = 31, int fastThreshold = 20) {
    return std::shared_ptr<ORB>(new ORB(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold));
  }
  void detectAndCompute(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors,
                        bool useProvidedKeypoints = false) {
    if (useProvidedKeypoints) throw std::runtime_error("dvslam::ORB: useProvidedKeypoints is not built");
    if (!mask.empty()) throw std::runtime_error("dvslam::ORB: detection masks are not built");
    if (image.empty()) return;                       // orb.cpp: returns before touching the outputs
    cv::Mat img = image.getMat();
    CV_Assert(img.type() == CV_8UC1);                // (cv::ORB converts colour images with cvtColor: dvs_bgr_to_gray does that here)
    std::vector<dvs_keypoint> kps;
    std::vector<uint8_t> desc;
    const int n = impl_.detectAndCompute(img.data, img.rows, img.cols, img.step, kps, desc);
    keypoints.resize(n);
    for (int i = 0; i < n; i++)
      keypoints[i] = cv::KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave, kps[i].class_id);
    if (n == 0) { descriptors.release(); return; }
    descriptors.create(n, 32, CV_8U);
    std::memcpy(descriptors.getMat().data, desc.data(), desc.size());
  }
  int descriptorSize() const { return 32; }

 private:
  ORB(int nf, float sf, int nl, int et, int fl, int wk, int st, int ps, int ft) : impl_(nf, sf, nl, et, fl, wk, st, ps, ft) {}
  CvOrb impl_;
};
#endif

}  // namespace dvslam
