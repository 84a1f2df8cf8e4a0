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

// ---- Keyframe.msg as rmw_fastrtps puts it on the wire (row N3).  The message is built as OBJECTS exactly like publishKeyframe
// does (frontend.cpp:699-776: per keypoint a Landmark and an Observation pushed onto two vectors), then serialised by a generic
// CDR stream that aligns on write (Fast-CDR, XCDR1: little endian, alignment origin = the byte after the 4-byte encapsulation
// header, strings = uint32 length including the NUL + bytes, sequences = uint32 count + elements, structs = their fields in
// declaration order).  Field order follows dynamic_visual_slam_interfaces/msg/{Keyframe,Landmark,Observation}.msg,
// std_msgs/Header (stamp{int32 sec, uint32 nanosec}, string frame_id), geometry_msgs/Transform (Vector3 translation,
// Quaternion rotation x y z w) and geometry_msgs/Point.  Fast-CDR / rosidl are absent from the image => parity unpinned.
}  // extern "C"  (C++ helpers)
#include <string>
#include <vector>
namespace {
struct CdrOut {
  std::vector<uint8_t> b;
  size_t origin = 0;
  void align(size_t a) { while ((b.size() - origin) % a) b.push_back(0); }
  template <class T> void put(T v) { align(sizeof(T)); const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + sizeof(T)); }
  void str(const std::string& s) { put<uint32_t>((uint32_t)s.size() + 1); b.insert(b.end(), s.begin(), s.end()); b.push_back(0); }
};
struct CdrIn {
  const uint8_t* p; size_t n, o = 0; bool ok = true;
  void align(size_t a) { o = (o + a - 1) / a * a; }
  template <class T> T get() { align(sizeof(T)); T v{}; if (o + sizeof(T) > n) { ok = false; return v; } memcpy(&v, p + o, sizeof(T)); o += sizeof(T); return v; }
};
struct LandmarkMsg { uint64_t landmark_id; double x, y, z; };
struct ObservationMsg { uint64_t landmark_id; double pixel_x, pixel_y; std::vector<uint8_t> descriptor; };
}  // namespace
extern "C" {

// returns the payload size (written to out if it fits in cap); *n_lm = landmarks in the message
size_t orc_publish_keyframe_cdr(int32_t sec, uint32_t nanosec, const char* header_frame_id, uint64_t keyframe_id, const double* trans,
                                const double* rot_xyzw, const orc_kp* kps, const uint8_t* desc, int n, const uint16_t* depth,
                                size_t step_bytes, float fx, float fy, float cx, float cy, const double* R, const double* t, uint8_t* out,
                                size_t cap, int32_t* n_lm) {
  std::vector<LandmarkMsg> landmarks;
  std::vector<ObservationMsg> observations;
  for (int i = 0; i < n; i++) {  // frontend.cpp:732-776
    const float px = kps[i].x, py = kps[i].y;
    const int x = static_cast<int>(std::round(px)), y = static_cast<int>(std::round(py));
    const float pt_depth = *(const uint16_t*)((const uint8_t*)depth + (size_t)y * step_bytes + 2 * (size_t)x) * 0.001f;
    const float X = (px - cx) * pt_depth / fx, Y = (py - cy) * pt_depth / fy, Z = pt_depth;
    if (Z > 0.3 && Z < 3.0) {
      const double v[3] = {X, Y, Z};
      double w[3];
      for (int r = 0; r < 3; r++) w[r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2]) + t[r];
      landmarks.push_back(LandmarkMsg{(uint64_t)i, w[0], w[1], w[2]});
      ObservationMsg o{(uint64_t)i, (double)px, (double)py, {}};
      o.descriptor.assign(desc + 32 * (size_t)i, desc + 32 * (size_t)i + 32);
      observations.push_back(o);
    }
  }
  CdrOut c;
  c.b = {0x00, 0x01, 0x00, 0x00};  // CDR_LE, no options
  c.origin = 4;
  c.put<int32_t>(sec); c.put<uint32_t>(nanosec); c.str(header_frame_id);           // std_msgs/Header
  c.put<uint64_t>(keyframe_id);                                                     // uint64 frame_id
  for (int k = 0; k < 3; k++) c.put<double>(trans[k]);                              // geometry_msgs/Transform
  for (int k = 0; k < 4; k++) c.put<double>(rot_xyzw[k]);
  c.put<uint32_t>((uint32_t)landmarks.size());                                      // Landmark[] landmarks
  for (const LandmarkMsg& l : landmarks) { c.put<uint64_t>(l.landmark_id); c.put<double>(l.x); c.put<double>(l.y); c.put<double>(l.z); }
  c.put<uint32_t>((uint32_t)observations.size());                                   // Observation[] observations
  for (const ObservationMsg& o : observations) {
    c.put<uint64_t>(o.landmark_id); c.put<double>(o.pixel_x); c.put<double>(o.pixel_y);
    c.put<uint32_t>((uint32_t)o.descriptor.size());
    c.b.insert(c.b.end(), o.descriptor.begin(), o.descriptor.end());
  }
  if (n_lm) *n_lm = (int32_t)landmarks.size();
  if (c.b.size() <= cap && out) memcpy(out, c.b.data(), c.b.size());
  return c.b.size();
}

// generic reader: returns 0 on success; arrays hold cap_n entries; descriptors of any length are accepted but only 32-byte ones copied
int orc_unpack_keyframe_cdr(const uint8_t* buf, size_t len, int32_t* sec, uint32_t* nanosec, char* frame_id, size_t frame_id_cap,
                            uint64_t* keyframe_id, double* trans, double* rot_xyzw, uint64_t* lm_ids, double* lm_xyz, uint64_t* obs_ids,
                            double* obs_px, uint8_t* obs_desc, int cap_n, int32_t* n_lm, int32_t* n_obs) {
  if (len < 4 || buf[0] != 0 || buf[1] != 1) return -1;
  CdrIn c{buf + 4, len - 4};
  *sec = c.get<int32_t>(); *nanosec = c.get<uint32_t>();
  const uint32_t sl = c.get<uint32_t>();
  if (!c.ok || sl == 0 || c.o + sl > c.n || sl > frame_id_cap) return -2;
  memcpy(frame_id, c.p + c.o, sl); c.o += sl;
  *keyframe_id = c.get<uint64_t>();
  for (int k = 0; k < 3; k++) trans[k] = c.get<double>();
  for (int k = 0; k < 4; k++) rot_xyzw[k] = c.get<double>();
  const uint32_t nl = c.get<uint32_t>();
  if (!c.ok || (int64_t)nl > cap_n) return -3;
  for (uint32_t i = 0; i < nl; i++) { lm_ids[i] = c.get<uint64_t>(); for (int k = 0; k < 3; k++) lm_xyz[3 * (size_t)i + k] = c.get<double>(); }
  const uint32_t no = c.get<uint32_t>();
  if (!c.ok || (int64_t)no > cap_n) return -4;
  for (uint32_t i = 0; i < no; i++) {
    obs_ids[i] = c.get<uint64_t>(); obs_px[2 * (size_t)i] = c.get<double>(); obs_px[2 * (size_t)i + 1] = c.get<double>();
    const uint32_t dl = c.get<uint32_t>();
    if (!c.ok || c.o + dl > c.n) return -5;
    if (dl == 32) memcpy(obs_desc + 32 * (size_t)i, c.p + c.o, 32);
    c.o += dl;
  }
  if (!c.ok) return -6;
  *n_lm = (int32_t)nl; *n_obs = (int32_t)no;
  return 0;
}

// cv::ORB's HarrisResponses (OpenCV features2d/src/orb.cpp, un-vendored; restated from the published source) for one layer:
// the offsets table, the integer gradient sums and the float response expression in the source's evaluation order.
void orc_harris_responses(const uint8_t* img, int rows, int cols, size_t stepb, const int32_t* xs, const int32_t* ys, int n, int blockSize,
                          float harris_k, float* out) {
  const uint8_t* ptr00 = img;
  const int step = (int)stepb;
  const int r = blockSize / 2;
  const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  std::vector<int> ofs(blockSize * blockSize);
  for (int i = 0; i < blockSize; i++)
    for (int j = 0; j < blockSize; j++) ofs[i * blockSize + j] = (int)(i * step + j);
  for (int ptidx = 0; ptidx < n; ptidx++) {
    const int x0 = xs[ptidx], y0 = ys[ptidx];
    if (!(x0 - r - 1 >= 0 && y0 - r - 1 >= 0 && x0 - r + blockSize <= cols - 1 && y0 - r + blockSize <= rows - 1)) { out[ptidx] = 0.f; continue; }
    const uint8_t* ptr0 = ptr00 + (y0 - r) * step + x0 - r;
    int a = 0, b = 0, c = 0;
    for (int k = 0; k < blockSize * blockSize; k++) {
      const uint8_t* ptr = ptr0 + ofs[k];
      const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
      const int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
    out[ptidx] = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
  }
}
}
