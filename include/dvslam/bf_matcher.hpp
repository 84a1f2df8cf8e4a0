// dvslam/bf_matcher.hpp — adapter with the call surface of cv::BFMatcher(cv::NORM_HAMMING, false).match as the
// reference uses it (src/frontend.cpp:220,614,1123; src/backend.cpp:222,1072) over the C-ABI.
//   dvslam::BFMatcher::match(q, nq, t, nt, matches)      plain pointers + dvslam::DMatch
//   (DVSLAM_WITH_OPENCV) match(const cv::Mat&, const cv::Mat&, std::vector<cv::DMatch>&)
// plus matchBelow() for the backend's association loop (backend.cpp:1068-1077): all (query, train) pairs with
// distance < max_dist in one launch instead of N_obs x N_landmarks 1x1 match() calls.
#pragma once
#include <climits>
#include <stdexcept>
#include <string>
#include <vector>
#include "../dvslam_hip.h"
#ifdef DVSLAM_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#endif

namespace dvslam {

struct DMatch {  // cv::DMatch layout
  int queryIdx, trainIdx, imgIdx;
  float distance;
};

class BFMatcher {
 public:
  explicit BFMatcher(int device = 0) {
    if (dvs_matcher_create(device, &m_) != DVS_OK) throw std::runtime_error(std::string("dvs_matcher_create: ") + dvs_last_error());
  }
  ~BFMatcher() { dvs_matcher_destroy(m_); }
  BFMatcher(const BFMatcher&) = delete;
  BFMatcher& operator=(const BFMatcher&) = delete;

  // one DMatch per query row in query order; empty train set -> empty result (cv behaviour)
  void match(const uint8_t* query, int nq, const uint8_t* train, int nt, std::vector<DMatch>& matches) const {
    matches.clear();
    if (nq <= 0 || nt <= 0) return;
    std::vector<int32_t> idx(nq), dist(nq);
    if (dvs_match_hamming(m_, query, nq, train, nt, idx.data(), dist.data()) != DVS_OK) throw std::runtime_error(dvs_last_error());
    matches.resize(nq);
    for (int i = 0; i < nq; i++) matches[i] = DMatch{i, idx[i], 0, (float)dist[i]};
  }
  // (queryIdx, trainIdx, distance) for every pair with distance < max_dist, query-major order
  void matchBelow(const uint8_t* query, int nq, const uint8_t* train, int nt, int max_dist, std::vector<DMatch>& out) const {
    out.clear();
    if (nq <= 0 || nt <= 0) return;
    int32_t n = 0;
    std::vector<int32_t> pairs(3 * 1024);
    if (dvs_match_hamming_thresh(m_, query, nq, train, nt, max_dist, pairs.data(), 1024, &n) != DVS_OK) throw std::runtime_error(dvs_last_error());
    if (n > 1024) {
      pairs.resize(3 * (size_t)n);
      if (dvs_match_hamming_thresh(m_, query, nq, train, nt, max_dist, pairs.data(), n, &n) != DVS_OK) throw std::runtime_error(dvs_last_error());
    }
    out.resize(n);
    for (int i = 0; i < n; i++) out[i] = DMatch{pairs[3 * i], pairs[3 * i + 1], 0, (float)pairs[3 * i + 2]};
  }
#ifdef DVSLAM_WITH_OPENCV
  void match(const cv::Mat& query, const cv::Mat& train, std::vector<cv::DMatch>& matches) const {
    CV_Assert(query.empty() || (query.type() == CV_8U && query.cols == 32 && query.isContinuous()));
    CV_Assert(train.empty() || (train.type() == CV_8U && train.cols == 32 && train.isContinuous()));
    std::vector<DMatch> m;
    match(query.data, query.rows, train.data, train.rows, m);
    matches.resize(m.size());
    for (size_t i = 0; i < m.size(); i++) matches[i] = cv::DMatch(m[i].queryIdx, m[i].trainIdx, 0, m[i].distance);
  }
#endif
  dvs_matcher* handle() { return m_; }

 private:
  dvs_matcher* m_ = nullptr;
};

#ifdef DVSLAM_WITH_OPENCV
// cv::BFMatcher's own constructor signature for the reference's two member declarations (frontend.cpp:220
// `matcher_(cv::NORM_HAMMING)`, backend.cpp:222 `descriptor_matcher_(cv::NORM_HAMMING, false)`): changing the member's TYPE
// from cv::BFMatcher to dvslam::HammingBFMatcher is the whole matcher-side integration; every match() call compiles unchanged.
class HammingBFMatcher : public BFMatcher {
 public:
  explicit HammingBFMatcher(int normType = cv::NORM_HAMMING, bool crossCheck = false) : BFMatcher(0) {
    CV_Assert(normType == cv::NORM_HAMMING && !crossCheck);  // the only configuration the reference uses
  }
};
#endif

}  // namespace dvslam
