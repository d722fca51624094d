// oracle/refkit: stand-in for <opencv2/highgui/highgui.hpp> (see ../../refkit.h): an 8-bit
// raster view with the element access the mosaic loop uses.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_
#define ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

typedef unsigned char uchar;

namespace cv {

struct Vec3b {
  uchar val[3];
  const uchar& operator[](int k) const { return val[k]; }
};

// rows x cols pixels of 1 or 3 bytes, rows `step` bytes apart, memory owned by the caller
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
  int rows, cols;
  const uint8_t* data;
  size_t step;
};

}  // namespace cv

#endif  // ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_
