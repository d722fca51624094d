// kindr_aslam_cv_lite.h -- stand-ins for the three remaining external types
// in the hot path's signatures (absent from this image):
//   kindr::minimal::QuatTransformation  (Pose,  aerial-mapper-io.h:17)
//   aslam::NCamera / aslam::Camera      (ortho-backward-grid.h:46,
//                                        ortho-backward-grid.cc:46,232)
//   cv::Mat                             (Image, aerial-mapper-io.h:19)
// Only the accessors the shim needs to read the data out are provided.
#ifndef AERIAL_MAPPER_COMPAT_KINDR_ASLAM_CV_LITE_H_
#define AERIAL_MAPPER_COMPAT_KINDR_ASLAM_CV_LITE_H_

#include <cstdint>
#include <memory>
#include <vector>

#include "aerial-mapper-compat/eigen_lite.h"

namespace kindr {
namespace minimal {

class RotationQuaternion {
 public:
  RotationQuaternion() {}
  RotationQuaternion(double w, double x, double y, double z) : q_(w, x, y, z) {}
  const Eigen::Quaterniond& toImplementation() const { return q_; }
  double w() const { return q_.w(); }
  double x() const { return q_.x(); }
  double y() const { return q_.y(); }
  double z() const { return q_.z(); }

 private:
  Eigen::Quaterniond q_;
};

class QuatTransformation {
 public:
  QuatTransformation() {}
  QuatTransformation(const RotationQuaternion& q, const Eigen::Vector3d& t) : q_(q), t_(t) {}
  const Eigen::Vector3d& getPosition() const { return t_; }
  const RotationQuaternion& getRotation() const { return q_; }

 private:
  RotationQuaternion q_;
  Eigen::Vector3d t_;
};

}  // namespace minimal
}  // namespace kindr

namespace aslam {

typedef kindr::minimal::QuatTransformation Transformation;

class Distortion {
 public:
  // (aslam_cv2 also has kFisheye: one such value here, so that the drop-in's refusal of models
  // it does not implement can be tested)
  enum class Type { kNoDistortion = 0, kRadTan = 1, kEquidistant = 2, kFisheye = 3 };
  Distortion() : type_(Type::kNoDistortion), params_(4) {}
  Distortion(Type type, double a, double b, double c, double d) : type_(type), params_(4) {
    params_(0) = a;
    params_(1) = b;
    params_(2) = c;
    params_(3) = d;
  }
  Type getType() const { return type_; }
  const Eigen::VectorXd& getParameters() const { return params_; }

 private:
  Type type_;
  Eigen::VectorXd params_;
};

// aslam::PinholeCamera: parameters = (fu, fv, cu, cv)
class Camera {
 public:
  enum class Type { kPinhole = 0, kUnifiedProjection = 1 };
  Camera(double fu, double fv, double cu, double cv, uint32_t width, uint32_t height,
         const Distortion& distortion = Distortion(), Type type = Type::kPinhole)
      : params_(4), width_(width), height_(height), distortion_(distortion), type_(type) {
    params_(0) = fu;
    params_(1) = fv;
    params_(2) = cu;
    params_(3) = cv;
  }
  uint32_t imageWidth() const { return width_; }
  uint32_t imageHeight() const { return height_; }
  const Eigen::VectorXd& getParameters() const { return params_; }
  const Distortion& getDistortion() const { return distortion_; }
  Type getType() const { return type_; }

 private:
  Eigen::VectorXd params_;
  uint32_t width_, height_;
  Distortion distortion_;
  Type type_;
};

class NCamera {
 public:
  typedef std::shared_ptr<NCamera> Ptr;
  NCamera(const Camera& camera, const Transformation& T_C_B) : camera_(camera), T_C_B_(T_C_B) {}
  const Camera& getCamera(size_t) const { return camera_; }
  const Transformation& get_T_C_B(size_t) const { return T_C_B_; }

 private:
  Camera camera_;
  Transformation T_C_B_;
};

}  // namespace aslam

namespace cv {

// 8-bit raster, 1 or 3 interleaved channels, row step in bytes.
class Mat {
 public:
  Mat() : rows(0), cols(0), data(nullptr), step(0), channels_(1) {}
  Mat(int r, int c, int ch) : rows(r), cols(c), step(static_cast<size_t>(c) * ch), channels_(ch) {
    store_.reset(new std::vector<uint8_t>(static_cast<size_t>(r) * step));
    data = store_->data();
  }
  Mat(int r, int c, int ch, uint8_t* external, size_t step_bytes)
      : rows(r), cols(c), data(external), step(step_bytes), channels_(ch) {}
  int channels() const { return channels_; }
  bool empty() const { return data == nullptr; }
  template <typename T>
  T& at(int row, int col) {
    return *reinterpret_cast<T*>(data + static_cast<size_t>(row) * step +
                                 static_cast<size_t>(col) * sizeof(T));
  }
  int rows, cols;
  uint8_t* data;
  size_t step;

 private:
  int channels_;
  std::shared_ptr<std::vector<uint8_t> > store_;
};

}  // namespace cv

#endif  // AERIAL_MAPPER_COMPAT_KINDR_ASLAM_CV_LITE_H_
