// ============================================================================================
// oracle/cvorb_oracle.cpp — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (same status as orb_oracle.cpp).
//
// CPU restatement of cv::ORB::create(nfeatures)->detectAndCompute(image, noArray(), keypoints, descriptors) as the
// reference uses it at /root/reference/dynamic_visual_slam/test/test_dbow2_integration.cpp:19,38 (SURVEY.md §8f row
// N4: "cv::ORB-compatible mode").  OpenCV is an un-vendored dependency of the reference (CMakeLists.txt:14, not pinned;
// Ubuntu 24.04 ships 4.6.0) and cannot be built or imported here, so the algorithm is restated from the published
// OpenCV 4.x sources, function by function:
//   features2d/src/orb.cpp        ORB_Impl::detectAndCompute, computeKeyPoints, HarrisResponses, ICAngles,
//                                 computeOrbDescriptors (WTA_K = 2, patchSize = 31, firstLevel = 0)
//   features2d/src/keypoint.cpp   KeyPointsFilter::runByImageBorder, retainBest (std::nth_element + std::partition:
//                                 the REAL libstdc++ routines are called here — they decide the output order)
//   imgproc/src/resize.cpp        INTER_LINEAR_EXACT for 8UC1: resize_bitExact<uchar, interpolationLinear<uchar>>,
//                                 ufixedpoint16 (Q8.8) coefficients from softdouble (= IEEE double, no fusion) arithmetic
// plus the primitives shared with orb_oracle.cpp (FAST-9/16 + nonmax, fastAtan2, the intensity centroid, the steered
// BRIEF test, the 7x7 fixed-point Gaussian).  Used ONLY by tests/ as the checker of the dvs_cvorb_* entry points.
// ============================================================================================
#include "orb_primitives.h"

namespace {

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR_EXACT), 8UC1, scale != 2 (orb.cpp: every level from the previous one)
void resizeLinearExactU8(const u8* src, int sw, int sh, size_t sstep, u8* dst, int dw, int dh, size_t dstep) {
  struct Axis { std::vector<int> ofs; std::vector<uint16_t> c0, c1; int lo = 0, hi = 0; };
  auto build = [](int ssize, int dsize) {
    Axis A;
    A.ofs.resize(dsize); A.c0.resize(dsize); A.c1.resize(dsize);
    A.lo = 0; A.hi = dsize;                                         // interpolationLinear: minofst / maxofst
    const double inv_scale = (double)dsize / ssize;                 // resize(): inv_scale_x = (double)dsize.width / ssize.width
    const double scale = 1.0 / inv_scale;                           // softdouble::one() / softdouble(inv_scale)
    for (int val = 0; val < dsize; val++) {
      const double fval = scale * ((double)val + 0.5) - 0.5;        // two roundings (the file is compiled -ffp-contract=off)
      int ival = cvFloorD(fval);
      if (ival >= 0 && ssize > 1) {
        if (ival < ssize - 1) {
          A.c1[val] = (uint16_t)cvRoundD((fval - (double)ival) * 256.0);   // ufixedpoint16(softdouble): cvRound(v * (1 << 8))
          A.c0[val] = (uint16_t)(256 - A.c1[val]);                          // one() - coeffs[1]
        } else {
          ival = ssize - 2; A.c0[val] = 0; A.c1[val] = 256;
          A.hi = std::min(A.hi, val);
        }
      } else {
        A.lo = std::max(A.lo, val + 1);
        ival = 0; A.c0[val] = 256; A.c1[val] = 0;
      }
      A.ofs[val] = ival;
    }
    return A;
  };
  const Axis X = build(sw, dw), Y = build(sh, dh);
  // hlineResize<uchar, ufixedpoint16, 2>: left of lo the first pixel, right of hi the last, Q8.8 in 16 bits in between
  std::vector<uint16_t> r0(dw), r1(dw);
  auto hline = [&](int sy, std::vector<uint16_t>& D) {
    const u8* S = src + (size_t)sy * sstep;
    int dx = 0;
    for (; dx < X.lo; dx++) D[dx] = (uint16_t)(S[0] << 8);
    for (; dx < X.hi; dx++) D[dx] = (uint16_t)(X.c0[dx] * S[X.ofs[dx]] + X.c1[dx] * S[X.ofs[dx] + 1]);
    for (; dx < dw; dx++) D[dx] = (uint16_t)(S[sw - 1] << 8);
  };
  for (int dy = 0; dy < dh; dy++) {
    u8* D = dst + (size_t)dy * dstep;
    if (dy < Y.lo || dy >= Y.hi) {   // vlineSet: the first / last source row alone, rounded from Q8.8
      hline(dy < Y.lo ? 0 : sh - 1, r0);
      for (int x = 0; x < dw; x++) D[x] = (u8)((r0[x] + 128) >> 8);
      continue;
    }
    hline(Y.ofs[dy], r0); hline(Y.ofs[dy] + 1, r1);
    const uint32_t b0 = Y.c0[dy], b1 = Y.c1[dy];
    for (int x = 0; x < dw; x++) D[x] = (u8)((b0 * r0[x] + b1 * r1[x] + 32768u) >> 16);   // ufixedpoint32 -> uchar: round half up
  }
}

struct CvOrb {
  int nfeatures = 500; double scaleFactor = 1.2f; int nlevels = 8, edgeThreshold = 31, scoreType = 0 /* HARRIS_SCORE */, fastThreshold = 20;
  int gk[7] = {18, 34, 48, 56, 48, 34, 18};
  static const int patchSize = 31, halfPatchSize = 15, HARRIS_BLOCK = 7;
  std::vector<Image> pyr, blurred;
  std::vector<float> layerScale;

  static float getScale(int level, int firstLevel, double scaleFactor) { return (float)std::pow(scaleFactor, (double)(level - firstLevel)); }

  // KeyPointsFilter::retainBest (keypoint.cpp) with the real std::nth_element / std::partition
  static void retainBest(std::vector<KeyPoint>& keypoints, int n_points) {
    if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
      if (n_points == 0) { keypoints.clear(); return; }
      std::nth_element(keypoints.begin(), keypoints.begin() + n_points - 1, keypoints.end(),
                       [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
      const float ambiguous_response = keypoints[n_points - 1].response;
      auto new_end = std::partition(keypoints.begin() + n_points, keypoints.end(),
                                    [ambiguous_response](const KeyPoint& k) { return k.response >= ambiguous_response; });
      keypoints.resize(new_end - keypoints.begin());
    }
  }

  int run(const u8* img, int rows, int cols, size_t step, std::vector<KeyPoint>& keypoints, std::vector<u8>& desc) {
    keypoints.clear(); desc.clear();
    if (!img || rows <= 0 || cols <= 0) return 0;                    // _image.empty(): returns without touching the outputs
    // pyramid: level sizes from the float scale table, every level resized from the previous one (orb.cpp: prevImg = currImg)
    pyr.assign(nlevels, Image()); layerScale.assign(nlevels, 1.f);
    for (int level = 0; level < nlevels; level++) {
      const float scale = getScale(level, 0, scaleFactor);
      layerScale[level] = scale;
      const float inv_scale = 1.0f / scale;
      const int w = cvRoundF(cols * inv_scale), h = cvRoundF(rows * inv_scale);
      pyr[level].create(w, h);
      if (level == 0) for (int y = 0; y < rows; y++) memcpy(pyr[0].row(y), img + (size_t)y * step, cols);
      else if (w > 0 && h > 0)
        resizeLinearExactU8(pyr[level - 1].d.data(), pyr[level - 1].cols, pyr[level - 1].rows, pyr[level - 1].cols, pyr[level].d.data(), w, h, w);
    }
    // ---- computeKeyPoints
    std::vector<int> nfeaturesPerLevel(nlevels);
    const float factor = (float)(1.0 / scaleFactor);
    float ndesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      nfeaturesPerLevel[level] = cvRoundF(ndesiredFeaturesPerScale);
      sumFeatures += nfeaturesPerLevel[level];
      ndesiredFeaturesPerScale *= factor;
    }
    nfeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
    std::vector<int> umax(halfPatchSize + 2);
    int v, v0, vmax = cvFloorF(halfPatchSize * std::sqrt(2.f) / 2 + 1);
    const int vmin = cvCeilF(halfPatchSize * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt((double)halfPatchSize * halfPatchSize - v * v));
    for (v = halfPatchSize, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
    std::vector<KeyPoint> all;
    std::vector<int> counters(nlevels);
    std::vector<FastPt> fp;
    for (int level = 0; level < nlevels; level++) {
      const int featuresNum = nfeaturesPerLevel[level];
      const Image& im = pyr[level];
      std::vector<KeyPoint> kps;
      if (im.cols > 0 && im.rows > 0) fast9_16(im.d.data(), im.cols, im.rows, im.cols, fastThreshold, fp); else fp.clear();
      for (const FastPt& p : fp) kps.push_back(KeyPoint{(float)p.x, (float)p.y, 7.f, -1.f, (float)p.score, 0, -1});
      // KeyPointsFilter::runByImageBorder(keypoints, img.size(), edgeThreshold)
      if (edgeThreshold > 0) {
        if (im.rows <= edgeThreshold * 2 || im.cols <= edgeThreshold * 2) kps.clear();
        else {
          const float x0 = (float)edgeThreshold, y0 = (float)edgeThreshold, x1 = x0 + (float)(im.cols - 2 * edgeThreshold), y1 = y0 + (float)(im.rows - 2 * edgeThreshold);
          kps.erase(std::remove_if(kps.begin(), kps.end(), [&](const KeyPoint& k) { return !(x0 <= k.x && k.x < x1 && y0 <= k.y && k.y < y1); }), kps.end());
        }
      }
      retainBest(kps, scoreType == 0 ? 2 * featuresNum : featuresNum);
      counters[level] = (int)kps.size();
      const float sf = layerScale[level];
      for (KeyPoint& k : kps) { k.octave = level; k.size = patchSize * sf; }
      all.insert(all.end(), kps.begin(), kps.end());
    }
    if (all.empty()) return 0;
    if (scoreType == 0) {
      // HarrisResponses(imagePyramid, layerInfo, allKeypoints, 7, HARRIS_K)
      const int blockSize = HARRIS_BLOCK, r = blockSize / 2;
      const float harris_k = 0.04f;
      const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
      const float scale_sq_sq = scale * scale * scale * scale;
      for (KeyPoint& k : all) {
        const Image& im = pyr[k.octave];
        const int step_ = im.cols, x0 = cvRoundF(k.x), y0 = cvRoundF(k.y);
        const u8* ptr0 = im.d.data() + (y0 - r) * step_ + x0 - r;
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < blockSize; i++)
          for (int j = 0; j < blockSize; j++) {
            const u8* ptr = ptr0 + i * step_ + j;
            const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step_ + 1] - ptr[-step_ - 1]) + (ptr[step_ + 1] - ptr[step_ - 1]);
            const int Iy = (ptr[step_] - ptr[-step_]) * 2 + (ptr[step_ - 1] - ptr[-step_ - 1]) + (ptr[step_ + 1] - ptr[-step_ + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
          }
        k.response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
      }
      std::vector<KeyPoint> newAll;
      int offset = 0;
      for (int level = 0; level < nlevels; level++) {
        std::vector<KeyPoint> kps(all.begin() + offset, all.begin() + offset + counters[level]);
        offset += counters[level];
        retainBest(kps, nfeaturesPerLevel[level]);
        newAll.insert(newAll.end(), kps.begin(), kps.end());
      }
      std::swap(all, newAll);
    }
    // ICAngles on the un-blurred pyramid, then pt *= scale
    for (KeyPoint& k : all) k.angle = icAngle(pyr[k.octave].d.data(), pyr[k.octave].cols, k.x, k.y, umax);
    for (KeyPoint& k : all) { const float scale = layerScale[k.octave]; k.x *= scale; k.y *= scale; }
    // ---- descriptors: every level blurred in place (GaussianBlur 7x7, sigma 2, BORDER_REFLECT_101), then computeOrbDescriptors
    blurred.assign(nlevels, Image());
    for (int level = 0; level < nlevels; level++)
      if (pyr[level].cols > 0 && pyr[level].rows > 0) gaussBlur7(pyr[level], blurred[level], gk);
    desc.assign(all.size() * 32, 0);
    for (size_t j = 0; j < all.size(); j++) {
      const KeyPoint& kpt = all[j];
      const float scale = 1.f / layerScale[kpt.octave];
      orbDescriptor(kpt.x * scale, kpt.y * scale, kpt.angle, blurred[kpt.octave].d.data(), blurred[kpt.octave].cols, &desc[j * 32]);
    }
    keypoints = all;
    return (int)all.size();
  }
};

}  // namespace

extern "C" {

struct orc_cv_keypoint { float x, y, size, angle, response; int32_t octave, class_id; };

void* orc_cvorb_create(int nfeatures, float scaleFactor, int nlevels, int edgeThreshold, int scoreType, int fastThreshold) {
  CvOrb* o = new CvOrb();
  o->nfeatures = nfeatures; o->scaleFactor = scaleFactor; o->nlevels = nlevels; o->edgeThreshold = edgeThreshold; o->scoreType = scoreType;
  o->fastThreshold = fastThreshold;
  return o;
}
void orc_cvorb_destroy(void* h) { delete (CvOrb*)h; }
void orc_cvorb_set_gauss_kernel(void* h, const int* k7) { memcpy(((CvOrb*)h)->gk, k7, sizeof(int) * 7); }
// returns the keypoint count (rows of the descriptor matrix); -3 when cap is too small
int orc_cvorb_detect_and_compute(void* h, const uint8_t* img, int rows, int cols, size_t step, orc_cv_keypoint* kps, uint8_t* desc, int cap) {
  std::vector<KeyPoint> k; std::vector<u8> d;
  const int n = ((CvOrb*)h)->run(img, rows, cols, step, k, d);
  if (n > cap) return -3;
  static_assert(sizeof(orc_cv_keypoint) == sizeof(KeyPoint), "layout");
  if (n) { memcpy(kps, k.data(), sizeof(KeyPoint) * n); memcpy(desc, d.data(), (size_t)n * 32); }
  return n;
}
int orc_cvorb_level(void* h, int level, int blurredFlag, uint8_t* dst, int cap, int* w, int* hh) {
  CvOrb* o = (CvOrb*)h;
  if (level < 0 || level >= (int)o->pyr.size()) return -6;
  const Image& im = blurredFlag ? o->blurred[level] : o->pyr[level];
  *w = im.cols; *hh = im.rows;
  if ((int)im.d.size() > cap) return -3;
  if (!im.d.empty()) memcpy(dst, im.d.data(), im.d.size());
  return 0;
}
void orc_resize_linear_exact_u8(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh, size_t dstep) {
  resizeLinearExactU8(src, sw, sh, sstep, dst, dw, dh, dstep);
}
// KeyPointsFilter::retainBest on bare responses: perm_out[i] = original index of the i-th survivor, in libstdc++'s order
int orc_retain_best(const float* responses, int n, int n_points, int32_t* perm_out) {
  std::vector<KeyPoint> k(n);
  for (int i = 0; i < n; i++) { k[i] = KeyPoint{0, 0, 0, 0, responses[i], 0, i}; }
  CvOrb::retainBest(k, n_points);
  for (size_t i = 0; i < k.size(); i++) perm_out[i] = k[i].class_id;
  return (int)k.size();
}

}  // extern "C"
