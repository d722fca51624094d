// oracle/refkit: stand-in for the OpenCV core / imgproc / highgui names the reference's hot-path
// files mention (see ../../refkit.h): a typed 8-bit / 16-bit raster, the element access of
// the mosaic loops, and -- for ortho-forward-homography.cc -- the handful of operations that
// file is made of, each forwarding to the oracle's restatement of the OpenCV algorithm
// (../../../amo_cvlike.h; NOT pinned by this build).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_
#define ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../refkit.h"
#include "../../../amo_cvlike.h"
#include "../../../amo_rectify.h"

typedef unsigned char uchar;

#define CV_8U 0
#define CV_16S 3
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16SC3 CV_MAKETYPE(CV_16S, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_INTER_LINEAR 1
#define CV_FILLED (-1)
#define CV_RGB2GRAY 7
#define CV_GRAY2RGB 8

namespace cv {

enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { BORDER_CONSTANT = 0 };

template <typename T>
using Ptr = std::shared_ptr<T>;

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

struct Point {
  int x, y;
  Point() : x(0), y(0) {}
  Point(int x_, int y_) : x(x_), y(y_) {}
};

struct Point2f {
  float x, y;
  Point2f() : x(0.0f), y(0.0f) {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(const Point& tl, const Point& br) : x(tl.x), y(tl.y), width(br.x - tl.x), height(br.y - tl.y) {}
};

struct Scalar {
  double val[4];
  Scalar(double v0 = 0.0) : val{v0, 0.0, 0.0, 0.0} {}
  Scalar(double v0, double v1, double v2) : val{v0, v1, v2, 0.0} {}
};

// rows x cols elements of `type`, rows `step` bytes apart: either a view of the caller's
// memory (the drivers hand the test's buffers in) or an owned, dense raster.
class Mat {
 public:
  Mat() : rows(0), cols(0), data(nullptr), step(0), type_(CV_8UC1) {}
  Mat(int r, int c, const uint8_t* pixels, size_t step_bytes, int type = CV_8UC1)
      : rows(r), cols(c), data(const_cast<uint8_t*>(pixels)), step(step_bytes), type_(type) {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
    step = static_cast<size_t>(c) * elemSize();
    own_.reset(new std::vector<uint8_t>(static_cast<size_t>(r) * step, 0));
    data = own_->data();
  }
  Mat(int r, int c, int type, const Scalar& fill) : rows(r), cols(c), type_(type) {
    step = static_cast<size_t>(c) * elemSize();
    own_.reset(new std::vector<uint8_t>(static_cast<size_t>(r) * step, static_cast<uint8_t>(fill.val[0])));
    data = own_->data();
  }
  void create(int r, int c, int type) { *this = Mat(r, c, type); }
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return (type_ >> 3) + 1; }
  size_t elemSize() const {
    const int d = depth();
    return static_cast<size_t>(channels()) * (d == CV_8U ? 1 : d == CV_16S ? 2 : d == CV_32F ? 4 : 8);
  }
  bool empty() const { return data == nullptr; }
  Size size() const { return Size(cols, rows); }
  template <typename T>
  const T& at(int row, int col) const {
    return *reinterpret_cast<const T*>(data + static_cast<size_t>(row) * step +
                                       static_cast<size_t>(col) * sizeof(T));
  }
  template <typename T>
  T& at(int row, int col) {
    return *reinterpret_cast<T*>(data + static_cast<size_t>(row) * step +
                                 static_cast<size_t>(col) * sizeof(T));
  }
  template <typename T>
  const T* ptr(int row) const {
    return reinterpret_cast<const T*>(data + static_cast<size_t>(row) * step);
  }
  template <typename T>
  T* ptr(int row) {
    return reinterpret_cast<T*>(data + static_cast<size_t>(row) * step);
  }
  Mat clone() const {
    Mat m(rows, cols, type_);
    const size_t line = static_cast<size_t>(cols) * elemSize();
    for (int r = 0; r < rows; ++r) std::memcpy(m.data + r * m.step, data + r * step, line);
    return m;
  }
  // 8-bit <-> 16-bit signed, channels kept, saturating (all the reference needs)
  void convertTo(Mat& dst, int rtype) const {
    const int ch = channels(), ddepth = rtype & 7;
    Mat out(rows, cols, CV_MAKETYPE(ddepth, ch));
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < cols * ch; ++k) {
        const int v = depth() == CV_8U ? ptr<uint8_t>(r)[k] : ptr<int16_t>(r)[k];
        if (ddepth == CV_8U)
          out.ptr<uint8_t>(r)[k] = static_cast<uint8_t>(v < 0 ? 0 : v > 255 ? 255 : v);
        else
          out.ptr<int16_t>(r)[k] = static_cast<int16_t>(v);
      }
    dst = out;
  }
  // every channel of the elements where the 8-bit mask is non-zero
  Mat& setTo(const Scalar& value, const Mat& mask) {
    const int ch = channels();
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        if (!mask.ptr<uint8_t>(r)[c]) continue;
        for (int k = 0; k < ch; ++k) {
          if (depth() == CV_8U)
            ptr<uint8_t>(r)[c * ch + k] = static_cast<uint8_t>(value.val[0]);
          else
            ptr<int16_t>(r)[c * ch + k] = static_cast<int16_t>(value.val[0]);
        }
      }
    return *this;
  }
  int rows, cols;
  uint8_t* data;
  size_t step;

 private:
  int type_;
  std::shared_ptr<std::vector<uint8_t> > own_;
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

// `mat > s`: per channel 255 / 0, 8-bit
inline Mat operator>(const Mat& a, double s) {
  const int ch = a.channels();
  Mat out(a.rows, a.cols, CV_MAKETYPE(CV_8U, ch));
  for (int r = 0; r < a.rows; ++r)
    for (int k = 0; k < a.cols * ch; ++k) {
      const double v = a.depth() == CV_8U ? a.ptr<uint8_t>(r)[k] : a.ptr<int16_t>(r)[k];
      out.ptr<uint8_t>(r)[k] = v > s ? 255 : 0;
    }
  return out;
}

// `s - mat` on an 8-bit raster, saturating
inline Mat operator-(int s, const Mat& a) {
  const int ch = a.channels();
  Mat out(a.rows, a.cols, a.type());
  for (int r = 0; r < a.rows; ++r)
    for (int k = 0; k < a.cols * ch; ++k) {
      const int v = s - a.ptr<uint8_t>(r)[k];
      out.ptr<uint8_t>(r)[k] = static_cast<uint8_t>(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  return out;
}

// CV_GRAY2RGB (replicate) and CV_RGB2GRAY (fixed point 4899 / 9617 / 1868 >> 14) on 8-bit rasters
inline void cvtColor(const Mat& src, Mat& dst, int code) {
  Mat out(src.rows, src.cols, code == CV_GRAY2RGB ? CV_8UC3 : CV_8UC1);
  for (int r = 0; r < src.rows; ++r)
    for (int c = 0; c < src.cols; ++c) {
      if (code == CV_GRAY2RGB) {
        const uint8_t v = src.ptr<uint8_t>(r)[c];
        uint8_t* o = out.ptr<uint8_t>(r) + 3 * c;
        o[0] = o[1] = o[2] = v;
      } else {
        const uint8_t* p = src.ptr<uint8_t>(r) + 3 * c;
        out.ptr<uint8_t>(r)[c] =
            static_cast<uint8_t>((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + (1 << 13)) >> 14);
      }
    }
  dst = out;
}

inline Mat getPerspectiveTransform(const std::vector<Point2f>& src, const std::vector<Point2f>& dst) {
  float s[4][2], d[4][2];
  for (int k = 0; k < 4; ++k) {
    s[k][0] = src[k].x;
    s[k][1] = src[k].y;
    d[k][0] = dst[k].x;
    d[k][1] = dst[k].y;
  }
  Mat M(3, 3, CV_64F);
  if (!amo::get_perspective_transform(s, d, M.ptr<double>(0))) refkit::check_fail("getPerspectiveTransform");
  return M;
}

inline void warpPerspective(const Mat& src, Mat& dst, const Mat& M, Size dsize, int /*flags*/,
                            int /*borderMode*/) {
  const amo::Image8 in = {src.data, src.step, src.cols, src.rows, src.channels()};
  std::vector<uint8_t> out;
  if (!amo::warp_nearest(in, M.ptr<double>(0), dsize.width, dsize.height, &out))
    refkit::check_fail("warpPerspective: singular matrix");
  Mat o(dsize.height, dsize.width, src.type());
  std::memcpy(o.data, out.data(), out.size());
  dst = o;
}

// what the reference writes to its output file is what a driver reads back
inline Mat& last_written() {
  static Mat m;
  return m;
}
// cv::remap(8UC1, CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT) and cv::drawContours(filled) as
// the rectifier calls them: the oracle's adopted definitions (../../../amo_rectify.h).
inline void remap(const Mat& src, Mat& dst, const Mat& map_x, const Mat& map_y, int /*interp*/,
                  int /*border*/, const Scalar& /*value*/) {
  dst = Mat(map_x.rows, map_x.cols, CV_8UC1);
  for (int v = 0; v < dst.rows; ++v)
    for (int u = 0; u < dst.cols; ++u)
      dst.at<uchar>(v, u) = amo::rect::remap_bilinear(src.data, src.step, src.cols, src.rows,
                                                      map_x.at<float>(v, u), map_y.at<float>(v, u));
}
inline void drawContours(Mat& image, const std::vector<std::vector<Point> >& contours, int index,
                         const Scalar& color, int /*thickness: CV_FILLED*/, int /*line type*/) {
  const std::vector<Point>& c = contours[static_cast<size_t>(index)];
  if (c.size() != 4) return;
  int cx[4], cy[4];
  for (int k = 0; k < 4; ++k) {
    cx[k] = c[k].x;
    cy[k] = c[k].y;
  }
  for (int v = 0; v < image.rows; ++v)
    for (int u = 0; u < image.cols; ++u)
      if (amo::rect::in_quad(cx, cy, u, v)) image.at<uchar>(v, u) = static_cast<uchar>(color.val[0]);
}

inline bool imwrite(const std::string&, const Mat& image) {
  last_written() = image.clone();
  return true;
}
inline void imshow(const std::string&, const Mat&) {}
inline int waitKey(int) { return 0; }

}  // namespace cv

#endif  // ORACLE_REFKIT_OPENCV2_HIGHGUI_HPP_
