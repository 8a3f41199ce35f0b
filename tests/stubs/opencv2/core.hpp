// Minimal functional stand-in for the handful of OpenCV core types that
// include/spfe_extractor.hpp touches.  TEST-ONLY: the build image has no OpenCV;
// this lets the adaptor be compiled and run in tests.  A real consumer includes
// the real <opencv2/core.hpp>.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC1 0
#define CV_16SC1 3
#define CV_32FC1 5

namespace cv {

inline int elemSize_(int type) { return type == CV_8UC1 ? 1 : type == CV_16SC1 ? 2 : 4; }

class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char *data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void *ext) : rows(r), cols(c), data((unsigned char *)ext), step((size_t)c * elemSize_(type)), type_(type) {}
  // cv::Mat::create: "if the current array shape and the type match the new ones, return immediately" (whoever shares the
  // buffer); else new, UNINITIALISED storage
  void create(int r, int c, int type) {
    if (data && buf_ && rows == r && cols == c && type_ == type) return;
    rows = r; cols = c; type_ = type; step = (size_t)c * elemSize_(type);
    buf_ = std::shared_ptr<unsigned char[]>(new unsigned char[(size_t)r * step + 1]);
    data = buf_.get();
  }
  int type() const { return type_; }
  bool empty() const { return rows == 0 || cols == 0 || !data; }
  void copyTo(Mat &dst) const {
    if (dst.rows != rows || dst.cols != cols || dst.type_ != type_ || !dst.data) dst.create(rows, cols, type_);
    for (int y = 0; y < rows; ++y) std::memcpy(dst.data + y * dst.step, data + y * step, (size_t)cols * elemSize_(type_));
  }
  Mat clone() const { Mat m; copyTo(m); return m; }
  template <class T> T &at(int y, int x) { return *reinterpret_cast<T *>(data + y * step + x * sizeof(T)); }
 private:
  int type_ = 0;
  std::shared_ptr<unsigned char[]> buf_;
};

struct Point2f { float x = 0, y = 0; };

}  // namespace cv
inline int cvRound(double v) { return (int)__builtin_lrint(v); }
namespace cv {

class KeyPoint {
 public:
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : size(s), angle(a), response(r), octave(o), class_id(c) { pt.x = x; pt.y = y; }
};

struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0;
  DMatch() {}
  DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};

class _InputArray {
 public:
  _InputArray(const Mat &m) : m_(&m) {}
  bool empty() const { return m_->empty(); }
  Mat getMat() const { return *m_; }
 private:
  const Mat *m_;
};
typedef const _InputArray &InputArray;

class _OutputArray {
 public:
  _OutputArray(Mat &m) : m_(&m) {}
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  Mat &getMat() const { return *m_; }
 private:
  Mat *m_;
};
typedef const _OutputArray &OutputArray;

}  // namespace cv
