// ============================================================================================
// oracle/frontend_oracle.cpp — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of the glue either side of the hot path:
//   cv::cvtColor(BGR2GRAY) 8-bit            frontend.cpp:1084     (OpenCV, un-vendored: color_rgb RGB2Gray<uchar>)
//   isValidDepth / filterDepth              frontend.cpp:457-527  (restated literally)
//   distance < 50 filter                    frontend.cpp:618-623, 1126-1132
//   publishKeyframe back-projection         frontend.cpp:732-776
//   associateObservation / reprojectPoint   backend.cpp:1064-1173 (geometric gate on the Hamming candidates)
// OpenCV is absent from the image: the cvtColor fixed-point variant (15-bit coefficients of OpenCV 4.x vs the 14-bit
// ones of older releases) and cv::Mat's 3x3 double product order could not be checked => parity unpinned.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
// ============================================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

extern "C" {

// variant 0: OpenCV 4.x  (B*3735 + G*19235 + R*9798 + 16384) >> 15 ; variant 1: older (B*1868 + G*9617 + R*4899 + 8192) >> 14
void orc_bgr_to_gray(const uint8_t* bgr, int rows, int cols, size_t step, uint8_t* gray, size_t gstep, int variant) {
  for (int y = 0; y < rows; y++) {
    const uint8_t* s = bgr + (size_t)y * step;
    uint8_t* d = gray + (size_t)y * gstep;
    for (int x = 0; x < cols; x++, s += 3)
      d[x] = variant == 0 ? (uint8_t)((s[0] * 3735 + s[1] * 19235 + s[2] * 9798 + (1 << 14)) >> 15)
                          : (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
  }
}

struct orc_kp { float x, y, size, angle, response; int32_t octave, class_id; };

// filterDepth: keeps keypoints whose rounded pixel has depth_mm * 0.001f in [min_depth, max_depth]; order preserved.
int orc_filter_depth(const orc_kp* kps, const uint8_t* desc, int n, const uint16_t* depth, int rows, int cols, size_t step_bytes,
                     float min_depth, float max_depth, orc_kp* out_kps, uint8_t* out_desc, int32_t* out_index) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const int x = static_cast<int>(std::round(kps[i].x));
    const int y = static_cast<int>(std::round(kps[i].y));
    if (x < 0 || y < 0 || x >= cols || y >= rows) continue;
    const float d = *(const uint16_t*)((const uint8_t*)depth + (size_t)y * step_bytes + 2 * (size_t)x) * 0.001f;
    if (d < min_depth || d > max_depth || std::isnan(d) || std::isinf(d) || d < 0.0f) continue;
    out_kps[m] = kps[i];
    if (desc) memcpy(out_desc + (size_t)m * 32, desc + (size_t)i * 32, 32);
    if (out_index) out_index[m] = i;
    m++;
  }
  return m;
}

// matches with (float)distance < max_distance, as (queryIdx, trainIdx, distance) triplets in query order
int orc_filter_matches(const int32_t* train_idx, const int32_t* dist, int n, float max_distance, int32_t* out) {
  int m = 0;
  for (int i = 0; i < n; i++)
    if ((float)dist[i] < max_distance) { out[3 * m] = i; out[3 * m + 1] = train_idx[i]; out[3 * m + 2] = dist[i]; m++; }
  return m;
}

// publishKeyframe: depth back-projection (float), range gate (double compare), world = R * p + t (double, row-major R)
int orc_backproject(const orc_kp* kps, int n, const uint16_t* depth, size_t step_bytes, float fx, float fy, float cx, float cy,
                    const double* R, const double* t, double* world_xyz, int32_t* out_index) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float px = kps[i].x, py = kps[i].y;
    const int x = static_cast<int>(std::round(px)), y = static_cast<int>(std::round(py));
    const float pt_depth = *(const uint16_t*)((const uint8_t*)depth + (size_t)y * step_bytes + 2 * (size_t)x) * 0.001f;
    const float X = (px - cx) * pt_depth / fx, Y = (py - cy) * pt_depth / fy, Z = pt_depth;
    if (Z > 0.3 && Z < 3.0) {
      const double v[3] = {X, Y, Z};
      for (int r = 0; r < 3; r++) world_xyz[3 * m + r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2]) + t[r];
      out_index[m] = i;
      m++;
    }
  }
  return m;
}

// reprojectPoint + cv::norm(obs.pixel - reprojection) for explicit (observation, landmark) pairs
void orc_reproject_errors(const int32_t* pairs, int npairs, const float* obs_px, const float* lm_xyz, const double* R, const double* t,
                          double fx, double fy, double cx, double cy, double* err) {
  for (int p = 0; p < npairs; p++) {
    const float* o = obs_px + 2 * (size_t)pairs[2 * p];
    const float* l = lm_xyz + 3 * (size_t)pairs[2 * p + 1];
    const double d[3] = {(double)l[0] - t[0], (double)l[1] - t[1], (double)l[2] - t[2]};
    double c[3];
    for (int r = 0; r < 3; r++) c[r] = R[r] * d[0] + R[3 + r] * d[1] + R[6 + r] * d[2];  // R^T * (X - t)
    float u, v;
    if (c[2] <= 0) { u = -1; v = -1; }
    else { u = (float)(fx * c[0] / c[2] + cx); v = (float)(fy * c[1] / c[2] + cy); }
    const float dx = o[0] - u, dy = o[1] - v;
    err[p] = std::sqrt((double)dx * dx + (double)dy * dy);
  }
}

// associateObservation on a database snapshot: per observation the candidate (Hamming < max_desc) with the smallest
// reprojection error < max_reproj, first in landmark order on ties; -1 if none
void orc_associate(const uint8_t* obs_desc, const float* obs_px, int nobs, const uint8_t* lm_desc, const float* lm_xyz, int nlm,
                   const double* R, const double* t, double fx, double fy, double cx, double cy, double max_desc, double max_reproj,
                   int32_t* best) {
  for (int i = 0; i < nobs; i++) {
    int bl = -1; double be = std::numeric_limits<double>::max();
    uint64_t a[4]; memcpy(a, obs_desc + (size_t)i * 32, 32);
    for (int j = 0; j < nlm; j++) {
      uint64_t b[4]; memcpy(b, lm_desc + (size_t)j * 32, 32);
      const int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) + __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
      if (!((float)d < max_desc)) continue;
      int32_t pr[2] = {i, j}; double e;
      orc_reproject_errors(pr, 1, obs_px, lm_xyz, R, t, fx, fy, cx, cy, &e);
      if (e < max_reproj && e < be) { bl = j; be = e; }
    }
    best[i] = bl;
  }
}
}
