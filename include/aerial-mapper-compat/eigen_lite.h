// eigen_lite.h -- the handful of Eigen names the hot path's C++ API mentions,
// for builds where Eigen is not installed (this image).  Member names and
// semantics follow Eigen 3; only what dsm.h / ortho-backward-grid.h / the
// demos' call sites use is provided.  With the real dependencies present,
// include/aerial-mapper-deps.h picks the real headers instead.
#ifndef AERIAL_MAPPER_COMPAT_EIGEN_LITE_H_
#define AERIAL_MAPPER_COMPAT_EIGEN_LITE_H_

#include <cstddef>
#include <memory>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

template <typename T>
using aligned_allocator = std::allocator<T>;

template <typename T, int N>
struct FixedVec {
  T v[N];
  FixedVec() {
    for (int i = 0; i < N; ++i) v[i] = T();
  }
  FixedVec(T a, T b) {
    static_assert(N == 2, "2 components");
    v[0] = a;
    v[1] = b;
  }
  FixedVec(T a, T b, T c) {
    static_assert(N == 3, "3 components");
    v[0] = a;
    v[1] = b;
    v[2] = c;
  }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
  T& x() { return v[0]; }
  T& y() { return v[1]; }
  const T& x() const { return v[0]; }
  const T& y() const { return v[1]; }
  const T* data() const { return v; }
  T* data() { return v; }
};

typedef FixedVec<double, 3> Vector3d;  // 24 bytes, x,y,z contiguous
typedef FixedVec<double, 2> Vector2d;
typedef FixedVec<double, 2> Array2d;
typedef FixedVec<int, 2> Array2i;

struct VectorXd {
  std::vector<double> v;
  VectorXd() {}
  explicit VectorXd(size_t n) : v(n, 0.0) {}
  double& operator()(size_t i) { return v[i]; }
  const double& operator()(size_t i) const { return v[i]; }
  double& operator[](size_t i) { return v[i]; }
  const double& operator[](size_t i) const { return v[i]; }
  size_t size() const { return v.size(); }
};

struct Quaterniond {
  double w_, x_, y_, z_;
  Quaterniond() : w_(1), x_(0), y_(0), z_(0) {}
  Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
  double w() const { return w_; }
  double x() const { return x_; }
  double y() const { return y_; }
  double z() const { return z_; }
};

// Column-major float matrix (grid_map::Matrix).
struct MatrixXf {
  std::vector<float> v;
  long rows_, cols_;
  MatrixXf() : rows_(0), cols_(0) {}
  void resize(long r, long c) {
    rows_ = r;
    cols_ = c;
    v.assign(static_cast<size_t>(r) * static_cast<size_t>(c), 0.0f);
  }
  void setConstant(float x) { v.assign(v.size(), x); }
  long rows() const { return rows_; }
  long cols() const { return cols_; }
  long size() const { return rows_ * cols_; }
  float* data() { return v.data(); }
  const float* data() const { return v.data(); }
  float& operator()(long i, long j) { return v[static_cast<size_t>(i + j * rows_)]; }
  const float& operator()(long i, long j) const {
    return v[static_cast<size_t>(i + j * rows_)];
  }
};

}  // namespace Eigen

#endif  // AERIAL_MAPPER_COMPAT_EIGEN_LITE_H_
