// oracle/refkit: stand-in for <opencv2/highgui/highgui.hpp> (see ../../refkit.h): an 8-bit
// raster view with the element access the mosaic loop uses.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_
#define ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_

#include <cstddef>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;

namespace cv {

struct Vec3b {
  uchar val[3];
  const uchar& operator[](int k) const { return val[k]; }
};

struct Vec3f {
  float val[3];
  Vec3f() : val{0.0f, 0.0f, 0.0f} {}
  Vec3f(float a, float b, float c) : val{a, b, c} {}
};

struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};

// A view of rows x cols elements of any type, rows `step` bytes apart, memory owned by the
// caller (the drivers hand the test's buffers in; the non-const row pointer is for the
// reference's `float* p = disparity_map.ptr<float>(v)`, which only reads).
class Mat {
 public:
  Mat() : rows(0), cols(0), data(nullptr), step(0) {}
  Mat(int r, int c, const uint8_t* pixels, size_t step_bytes)
      : rows(r), cols(c), data(pixels), step(step_bytes) {}
  template <typename T>
  const T& at(int row, int col) const {
    return *reinterpret_cast<const T*>(data + static_cast<size_t>(row) * step +
                                       static_cast<size_t>(col) * sizeof(T));
  }
  template <typename T>
  const T* ptr(int row) const {
    return reinterpret_cast<const T*>(data + static_cast<size_t>(row) * step);
  }
  template <typename T>
  T* ptr(int row) {
    return const_cast<T*>(reinterpret_cast<const T*>(data + static_cast<size_t>(row) * step));
  }
  Size size() const { return Size(cols, rows); }
  int rows, cols;
  const uint8_t* data;
  size_t step;
};

// cv::Mat_<T>: an owned raster (the densifier's organised cloud, written but not read here)
template <typename T>
class Mat_ {
 public:
  void create(const Size& s) { v_.assign(static_cast<size_t>(s.width) * static_cast<size_t>(s.height), T()); }
  void setTo(const T& value) { v_.assign(v_.size(), value); }

 private:
  std::vector<T> v_;
};

}  // namespace cv

#endif  // ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_
