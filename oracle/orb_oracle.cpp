// ============================================================================================
// oracle/orb_oracle.cpp — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the reference's ORB hot path, used ONLY by tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg as the checker / timed CPU baseline.  Nothing under
// dynamic-visual-slam_amd/ may include, link or call this file.
//
// Reference (read-only, /root/reference/dynamic_visual_slam):
//   src/ORBextractor.cpp           (ORBX.cpp)  — control flow restated function by function below
//   include/.../ORBextractor.hpp   (ORBX.hpp)
// Third-party arithmetic that the reference calls but does not vendor (OpenCV, un-pinned by
// CMakeLists.txt:14; Ubuntu 24.04 / ROS 2 Jazzy ships 4.6.0) is restated from the published
// OpenCV 4.x algorithms:
//   cv::resize INTER_LINEAR 8UC1   (imgproc/resize.cpp: fixed-point 11-bit coefficients)
//   cv::FAST TYPE_9_16 + nonmax    (features2d/fast.cpp, fast_score.cpp)
//   cv::GaussianBlur 7x7 s=2 8UC1  (imgproc/smooth.dispatch.cpp fixed-point ufixedpoint16 path)
//   cv::fastAtan2, cvRound         (core/mathfuncs_core, fast_math.hpp)
// "PARITY UNPINNED": the reference ships no golden vectors or known-answer tests for this path
//  and OpenCV cannot be built or imported in the build container, so this
// restatement could not be checked against outputs of the reference itself.  It is pinned only
// by structural known-answers derivable from the reference text (level sizes, quotas, umax,
// pattern checksum) — tests/test_oracle_orb.py.
//
// libstdc++'s std::sort / std::list are used on purpose in the quad-tree: the reference's tie
// order comes from them (ORBX.cpp:700).
// ============================================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "orb_primitives.h"

namespace {

// ---- quad-tree  (ORBX.hpp:31-42, ORBX.cpp:480-779) --------------------------------------------
struct Pt2i { int x = 0, y = 0; };
struct ExtractorNode {
  std::vector<KeyPoint> vKeys;
  Pt2i UL, UR, BL, BR;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore = false;
  void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) {
  const int halfX = (int)ceil(static_cast<float>(UR.x - UL.x) / 2);
  const int halfY = (int)ceil(static_cast<float>(BR.y - UL.y) / 2);
  n1.UL = UL;                       n1.UR = {UL.x + halfX, UL.y};
  n1.BL = {UL.x, UL.y + halfY};     n1.BR = {UL.x + halfX, UL.y + halfY};
  n2.UL = n1.UR; n2.UR = UR; n2.BL = n1.BR; n2.BR = {UR.x, UL.y + halfY};
  n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = BL; n3.BR = {n1.BR.x, BL.y};
  n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = BR;
  for (size_t i = 0; i < vKeys.size(); i++) {
    const KeyPoint& kp = vKeys[i];
    if (kp.x < n1.UR.x) { if (kp.y < n1.BR.y) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp); }
    else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
    else n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

bool compareNodes(std::pair<int, ExtractorNode*>& e1, std::pair<int, ExtractorNode*>& e2) {
  if (e1.first < e2.first) return true;
  else if (e1.first > e2.first) return false;
  else return e1.second->UL.x < e2.second->UL.x;
}

std::vector<KeyPoint> distributeOctTree(const std::vector<KeyPoint>& keys, int minX, int maxX, int minY, int maxY,
                                        int N, int reserveHint) {
  const int nIni = (int)round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<ExtractorNode> lNodes;
  std::vector<ExtractorNode*> vpIniNodes(nIni);
  for (int i = 0; i < nIni; i++) {
    ExtractorNode ni;
    ni.UL = {(int)(hX * static_cast<float>(i)), 0};
    ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = {ni.UL.x, maxY - minY};
    ni.BR = {ni.UR.x, maxY - minY};
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < keys.size(); i++) vpIniNodes[(size_t)(keys[i].x / hX)]->vKeys.push_back(keys[i]);

  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
    else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
    else lit++;
  }
  bool bFinish = false;
  std::vector<std::pair<int, ExtractorNode*>> vSizeAndPointerToNode;
  auto addChild = [&](ExtractorNode& n, int* nToExpand) {
    if (n.vKeys.size() > 0) {
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) { lit++; continue; }
      ExtractorNode n1, n2, n3, n4;
      lit->DivideNode(n1, n2, n3, n4);
      addChild(n1, &nToExpand); addChild(n2, &nToExpand); addChild(n3, &nToExpand); addChild(n4, &nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<std::pair<int, ExtractorNode*>> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), compareNodes);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          ExtractorNode n1, n2, n3, n4;
          vPrev[j].second->DivideNode(n1, n2, n3, n4);
          addChild(n1, nullptr); addChild(n2, nullptr); addChild(n3, nullptr); addChild(n4, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  std::vector<KeyPoint> res;
  res.reserve(reserveHint);
  for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
    std::vector<KeyPoint>& v = it->vKeys;
    KeyPoint* p = &v[0];
    float maxResponse = p->response;
    for (size_t k = 1; k < v.size(); k++)
      if (v[k].response > maxResponse) { p = &v[k]; maxResponse = v[k].response; }
    res.push_back(*p);
  }
  return res;
}

// ---- extractor object (ORBX.cpp:409-469 ctor, 781-896, 1086-1194) -------------------------------
struct Extractor {
  int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;  // ORBX.hpp:96-100 (scaleFactor is double)
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<Image> pyr, blurred;
  std::vector<std::vector<KeyPoint>> cand;     // per level, region-relative, before the quad-tree
  std::vector<std::vector<KeyPoint>> levelKps; // per level, level coordinates, with angle
  int gk[7] = {18, 34, 48, 56, 48, 34, 18};    // OpenCV >= 4.5.2 getGaussianKernelFixedPoint_ED(7, 2.0)

  Extractor(int nf, float sf, int nl, int ini, int mn) : nfeatures(nf), scaleFactor(sf), nlevels(nl), iniThFAST(ini), minThFAST(mn) {
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cvRoundF(nDesired);
      sum += mnFeaturesPerLevel[level];
      nDesired *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloorF(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = cvCeilF(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  void levelSize(int cols, int rows, int level, int& w, int& h) const {
    float scale = mvInvScaleFactor[level];
    w = cvRoundF((float)cols * scale);
    h = cvRoundF((float)rows * scale);
  }

  // The reference divides by nCols / nIni without checking; those sizes are UB there.  The oracle
  // reports them as unsupported (-2) instead of reproducing UB.
  bool supported(int cols, int rows) const {
    for (int l = 0; l < nlevels; l++) {
      int w, h; levelSize(cols, rows, l, w, h);
      const float width = (float)((w - EDGE_THRESHOLD + 3) - (EDGE_THRESHOLD - 3));
      const float height = (float)((h - EDGE_THRESHOLD + 3) - (EDGE_THRESHOLD - 3));
      if ((int)(width / 35.f) < 1 || (int)(height / 35.f) < 1) return false;
      if ((int)round(width / height) < 1) return false;
    }
    return true;
  }

  void computePyramid(const u8* img, int rows, int cols, size_t step) {
    pyr.resize(nlevels);
    for (int level = 0; level < nlevels; ++level) {
      int w, h; levelSize(cols, rows, level, w, h);
      pyr[level].create(w, h);
      if (level != 0)
        resizeLinearU8(pyr[level - 1].d.data(), pyr[level - 1].cols, pyr[level - 1].rows, pyr[level - 1].cols,
                       pyr[level].d.data(), w, h, w);
      else
        for (int y = 0; y < rows; y++) memcpy(pyr[0].row(y), img + (size_t)y * step, cols);
      // copyMakeBorder (ORBX.cpp:1184-1190): the 19-px border is never read by later stages.
    }
  }

  void computeKeyPointsOctTree() {
    cand.assign(nlevels, {});
    levelKps.assign(nlevels, {});
    const float W = 35;
    for (int level = 0; level < nlevels; ++level) {
      const Image& im = pyr[level];
      const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
      const int maxBorderX = im.cols - EDGE_THRESHOLD + 3, maxBorderY = im.rows - EDGE_THRESHOLD + 3;
      std::vector<KeyPoint>& vToDistributeKeys = cand[level];
      const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
      const int nCols = (int)(width / W), nRows = (int)(height / W);
      const int wCell = (int)ceil(width / nCols), hCell = (int)ceil(height / nRows);
      std::vector<FastPt> cell;
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;  // rowRange/colRange(int,int)
          const u8* sub = im.row(y0) + x0;
          fast9_16(sub, x1 - x0, y1 - y0, im.cols, iniThFAST, cell);
          if (cell.empty()) fast9_16(sub, x1 - x0, y1 - y0, im.cols, minThFAST, cell);
          for (const FastPt& p : cell) {
            KeyPoint kp{(float)p.x, (float)p.y, 7.f, -1.f, (float)p.score, 0, -1};
            kp.x += j * wCell;
            kp.y += i * hCell;
            vToDistributeKeys.push_back(kp);
          }
        }
      }
      std::vector<KeyPoint>& keypoints = levelKps[level];
      keypoints = distributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                    mnFeaturesPerLevel[level], nfeatures);
      const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
      for (KeyPoint& kp : keypoints) {
        kp.x += minBorderX; kp.y += minBorderY; kp.octave = level; kp.size = (float)scaledPatchSize;
      }
    }
    for (int level = 0; level < nlevels; ++level)
      for (KeyPoint& kp : levelKps[level])
        kp.angle = icAngle(pyr[level].d.data(), pyr[level].cols, kp.x, kp.y, umax);
  }

  // operator() (ORBX.cpp:1086-1167) with vLappingArea = {0,0} as the frontend passes (FE:290)
  int run(const u8* img, int rows, int cols, size_t step, std::vector<KeyPoint>& out, std::vector<u8>& desc) {
    out.clear(); desc.clear();
    if (!img || rows <= 0 || cols <= 0) return -1;
    if (!supported(cols, rows)) return -2;
    computePyramid(img, rows, cols, step);
    computeKeyPointsOctTree();
    int nk = 0;
    for (int l = 0; l < nlevels; l++) nk += (int)levelKps[l].size();
    out.resize(nk); desc.assign((size_t)nk * 32, 0);
    blurred.assign(nlevels, Image());
    int monoIndex = 0, stereoIndex = nk - 1;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPoint> keypoints = levelKps[level];  // copy: levelKps keeps level coordinates
      if (keypoints.empty()) continue;
      gaussBlur7(pyr[level], blurred[level], gk);
      std::vector<u8> d((size_t)keypoints.size() * 32);
      for (size_t i = 0; i < keypoints.size(); i++)
        orbDescriptor(keypoints[i].x, keypoints[i].y, keypoints[i].angle, blurred[level].d.data(), blurred[level].cols, &d[i * 32]);
      float scale = mvScaleFactor[level];
      for (size_t i = 0; i < keypoints.size(); i++) {
        KeyPoint& kp = keypoints[i];
        if (level != 0) { kp.x *= scale; kp.y *= scale; }
        if (kp.x >= 0 && kp.x <= 0) {  // vLappingArea = {0,0}; never true (x >= 19)
          out[stereoIndex] = kp; memcpy(&desc[(size_t)stereoIndex * 32], &d[i * 32], 32); stereoIndex--;
        } else {
          out[monoIndex] = kp; memcpy(&desc[(size_t)monoIndex * 32], &d[i * 32], 32); monoIndex++;
        }
      }
    }
    return monoIndex;
  }
};

}  // namespace

// ================================= C entry points (ctypes / bench) =============================
extern "C" {

struct orc_keypoint { float x, y, size, angle, response; int32_t octave, class_id; };

void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void orc_orb_destroy(void* h) { delete (Extractor*)h; }
void orc_orb_set_gauss_kernel(void* h, const int* k7) { memcpy(((Extractor*)h)->gk, k7, sizeof(int) * 7); }

// returns number of keypoints (>=0), -1 for empty input, -2 unsupported size, -3 capacity too small
int orc_orb_extract(void* h, const uint8_t* img, int rows, int cols, size_t step, orc_keypoint* kps, uint8_t* desc, int cap) {
  Extractor* e = (Extractor*)h;
  std::vector<KeyPoint> out; std::vector<u8> d;
  int n = e->run(img, rows, cols, step, out, d);
  if (n < 0) return n;
  if ((int)out.size() > cap) return -3;
  static_assert(sizeof(orc_keypoint) == sizeof(KeyPoint), "layout");
  if (!out.empty()) { memcpy(kps, out.data(), out.size() * sizeof(KeyPoint)); memcpy(desc, d.data(), d.size()); }
  return (int)out.size();
}

void orc_orb_tables(void* h, float* scale, float* invScale, int* featPerLevel, int* umax16) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; i++) { scale[i] = e->mvScaleFactor[i]; invScale[i] = e->mvInvScaleFactor[i]; featPerLevel[i] = e->mnFeaturesPerLevel[i]; }
  for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}
void orc_orb_level_size(void* h, int cols, int rows, int level, int* w, int* hh) { ((Extractor*)h)->levelSize(cols, rows, level, *w, *hh); }

// stage outputs of the last orc_orb_extract call
int orc_orb_get_level(void* h, int level, int blurredFlag, uint8_t* dst, int cap) {
  Extractor* e = (Extractor*)h;
  const Image& im = blurredFlag ? e->blurred[level] : e->pyr[level];
  if ((int)im.d.size() > cap) return -3;
  if (!im.d.empty()) memcpy(dst, im.d.data(), im.d.size());
  return (int)im.d.size();
}
// candidates before the quad-tree: region-relative x,y and score, int32 triplets
int orc_orb_get_candidates(void* h, int level, int32_t* xys, int cap) {
  Extractor* e = (Extractor*)h;
  const auto& c = e->cand[level];
  if ((int)c.size() > cap) return -3;
  for (size_t i = 0; i < c.size(); i++) { xys[3 * i] = (int)c[i].x; xys[3 * i + 1] = (int)c[i].y; xys[3 * i + 2] = (int)c[i].response; }
  return (int)c.size();
}
int orc_orb_get_level_keypoints(void* h, int level, orc_keypoint* kps, int cap) {
  Extractor* e = (Extractor*)h;
  const auto& c = e->levelKps[level];
  if ((int)c.size() > cap) return -3;
  if (!c.empty()) memcpy(kps, c.data(), c.size() * sizeof(KeyPoint));
  return (int)c.size();
}

// stand-alone stage functions
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh, size_t dstep) {
  resizeLinearU8(src, sw, sh, sstep, dst, dw, dh, dstep);
}
int orc_fast(const uint8_t* img, int cols, int rows, size_t step, int threshold, int32_t* xys, int cap) {
  std::vector<FastPt> v; fast9_16(img, cols, rows, step, threshold, v);
  if ((int)v.size() > cap) return -3;
  for (size_t i = 0; i < v.size(); i++) { xys[3 * i] = v[i].x; xys[3 * i + 1] = v[i].y; xys[3 * i + 2] = v[i].score; }
  return (int)v.size();
}
void orc_gauss7(const uint8_t* src, int cols, int rows, uint8_t* dst, const int* k7) {
  Image s, d; s.create(cols, rows); memcpy(s.d.data(), src, (size_t)cols * rows);
  gaussBlur7(s, d, k7); memcpy(dst, d.d.data(), (size_t)cols * rows);
}
float orc_fast_atan2(float y, float x) { return fastAtan2f(y, x); }
float orc_ic_angle(const uint8_t* img, size_t step, float x, float y) {
  static Extractor e(1000, 1.2f, 8, 20, 7);
  return icAngle(img, step, x, y, e.umax);
}
void orc_descriptor(const uint8_t* blurredImg, size_t step, float x, float y, float angle, uint8_t* desc32) {
  orbDescriptor(x, y, angle, blurredImg, step, desc32);
}
// quad-tree alone: xys = region-relative candidates (int triplets) in candidate order
int orc_distribute(const int32_t* xys, int n, int minX, int maxX, int minY, int maxY, int N, int32_t* out_xys, int cap) {
  std::vector<KeyPoint> in(n);
  for (int i = 0; i < n; i++) in[i] = KeyPoint{(float)xys[3 * i], (float)xys[3 * i + 1], 7.f, -1.f, (float)xys[3 * i + 2], 0, -1};
  std::vector<KeyPoint> r = distributeOctTree(in, minX, maxX, minY, maxY, N, N);
  if ((int)r.size() > cap) return -3;
  for (size_t i = 0; i < r.size(); i++) { out_xys[3 * i] = (int)r[i].x; out_xys[3 * i + 1] = (int)r[i].y; out_xys[3 * i + 2] = (int)r[i].response; }
  return (int)r.size();
}
// std::sort with the reference comparator on (count, ULx) pairs; payload = original index. Used to
// check the product's libstdc++-introsort replica.
void orc_std_sort_nodes(const int32_t* count, const int32_t* ulx, int n, int32_t* perm) {
  std::vector<ExtractorNode> nodes(n);
  std::vector<std::pair<int, ExtractorNode*>> v(n);
  for (int i = 0; i < n; i++) { nodes[i].UL.x = ulx[i]; v[i] = {count[i], &nodes[i]}; }
  std::sort(v.begin(), v.end(), compareNodes);
  for (int i = 0; i < n; i++) perm[i] = (int)(v[i].second - nodes.data());
}
const int8_t* orc_brief_pattern() { return kPattern; }
void orc_sincosf(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }

}  // extern "C"
