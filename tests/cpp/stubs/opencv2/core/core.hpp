// TEST-ONLY stand-in for the handful of OpenCV core types the DVSLAM_WITH_OPENCV adapters touch (cv::Mat, InputArray /
// OutputArray, KeyPoint, DMatch, Point3d, CV_Assert).  It exists so that the adapter branches a maintainer compiles against
// the real OpenCV are at least COMPILED (and their data movement exercised) in an image that has no OpenCV.  It pins nothing
// about OpenCV's arithmetic, is never linked into libdvslam_hip.so, and must not be used outside tests/.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

typedef unsigned char uchar;   // as opencv2/core/hal/interface.h declares it, in the global namespace
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

namespace cv {
enum { NORM_HAMMING = 6 };

class Mat {
 public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t step_bytes = 0) : rows(r), cols(c), data((uint8_t*)ext), type_(type) {
    step = step_bytes ? step_bytes : (size_t)c * esz();
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && buf_) return;
    rows = r; cols = c; type_ = type; step = (size_t)c * esz();
    buf_ = std::make_shared<std::vector<uint8_t>>((size_t)r * step);
    data = buf_->data();
  }
  void release() { buf_.reset(); data = nullptr; rows = cols = 0; step = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  size_t total() const { return (size_t)rows * cols; }
  int type() const { return type_; }
  bool isContinuous() const { return step == (size_t)cols * esz(); }
  Mat clone() const {
    Mat m;
    if (empty()) return m;
    m.create(rows, cols, type_);
    for (int i = 0; i < rows; i++) std::memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols * esz());
    return m;
  }
  template <class T> T& at(int i, int j) { return *(T*)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
  template <class T> const T& at(int i, int j) const { return *(const T*)(data + (size_t)i * step + (size_t)j * sizeof(T)); }
  template <class T> T& at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
  template <class T> const T& at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }

 private:
  size_t esz() const { return type_ == CV_64F ? 8 : (type_ == CV_32F ? 4 : 1); }
  int type_ = CV_8U;
  std::shared_ptr<std::vector<uint8_t>> buf_;
};

class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
  bool empty() const { return !m_ || m_->empty(); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 protected:
  Mat* m_ = nullptr;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) { m_ = &m; }
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline const _InputArray& noArray() { static _InputArray none; return none; }

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point3d { double x = 0, y = 0, z = 0; Point3d() {} Point3d(double a, double b, double c) : x(a), y(b), z(c) {} };
struct Point3f { float x = 0, y = 0, z = 0; Point3f() {} Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };
struct KeyPoint {
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0;
  DMatch() {}
  DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};
}  // namespace cv
