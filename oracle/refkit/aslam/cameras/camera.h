// oracle/refkit: stand-in for aslam_cv2's camera / ncamera headers and minkindr's
// QuatTransformation (see ../../refkit.h).  The arithmetic of project3 and of the
// transformation's inverse / transform / product is the oracle's adopted definition
// (../../../amo_compat.h) -- NOT pinned by this build.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_ASLAM_CAMERAS_CAMERA_H_
#define ORACLE_REFKIT_ASLAM_CAMERAS_CAMERA_H_

#include <cstdint>
#include <memory>
#include <vector>

#include <Eigen/Core>
#include <glog/logging.h>  // (the real aslam headers bring glog in; the reference relies on it)

#include "../../../amo_cvlike.h"
#include "distortion.h"

namespace kindr {
namespace minimal {

class RotationQuaternion {
 public:
  RotationQuaternion(double w, double x, double y, double z) : q_(w, x, y, z) {}
  const Eigen::Quaterniond& toImplementation() const { return q_; }

 private:
  Eigen::Quaterniond q_;
};

class QuatTransformation {
 public:
  QuatTransformation() {
    const double identity[7] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};
    p_ = amo::pose_from7(identity);
  }
  explicit QuatTransformation(const amo::Pose& p) : p_(p) {}
  QuatTransformation inverse() const { return QuatTransformation(amo::inverse(p_)); }
  Eigen::Vector3d transform(const Eigen::Vector3d& v) const {
    const amo::Vec3 in = {v(0), v(1), v(2)};
    const amo::Vec3 out = amo::transform(p_, in);
    return Eigen::Vector3d(out.x, out.y, out.z);
  }
  QuatTransformation operator*(const QuatTransformation& rhs) const {
    return QuatTransformation(amo::compose(p_, rhs.p_));
  }
  Eigen::Vector3d getPosition() const { return Eigen::Vector3d(p_.t.x, p_.t.y, p_.t.z); }
  Eigen::Matrix3d getRotationMatrix() const {
    double R[9];
    amo::rotation_matrix(p_.q, R);
    Eigen::Matrix3d m;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j];
    return m;
  }
  const amo::Pose& pose() const { return p_; }
  // (minkindr: a RotationQuaternion(w, x, y, z) + a position; getRotation().toImplementation()
  // is the Eigen quaternion)
  QuatTransformation(const RotationQuaternion& q, const Eigen::Vector3d& t) {
    const double p[7] = {t(0), t(1), t(2), q.toImplementation().w(), q.toImplementation().x(),
                         q.toImplementation().y(), q.toImplementation().z()};
    p_ = amo::pose_from7(p);
  }
  RotationQuaternion getRotation() const {
    return RotationQuaternion(p_.q.w, p_.q.x, p_.q.y, p_.q.z);
  }

 private:
  amo::Pose p_;
};

}  // namespace minimal
}  // namespace kindr

namespace aslam {

typedef kindr::minimal::QuatTransformation Transformation;

class ProjectionResult {
 public:
  enum Status {
    KEYPOINT_VISIBLE = amo::KEYPOINT_VISIBLE,
    KEYPOINT_OUTSIDE_IMAGE_BOX = amo::KEYPOINT_OUTSIDE_IMAGE_BOX,
    POINT_BEHIND_CAMERA = amo::POINT_BEHIND_CAMERA,
    PROJECTION_INVALID = amo::PROJECTION_INVALID
  };
  explicit ProjectionResult(Status s) : status_(s) {}
  Status getDetailedStatus() const { return status_; }

 private:
  Status status_;
};

class Camera {
 public:
  explicit Camera(const amo_camera& c) : c_(c) {}
  ProjectionResult project3(const Eigen::Vector3d& point, Eigen::Vector2d* keypoint) const {
    const amo::Vec3 p = {point(0), point(1), point(2)};
    const amo::ProjectionStatus st = amo::project3(c_, p, &(*keypoint)(0), &(*keypoint)(1));
    return ProjectionResult(static_cast<ProjectionResult::Status>(st));
  }
  // PinholeCamera::backProject3: the ray of a pixel, z = 1 (distortion undone)
  bool backProject3(const Eigen::Vector2d& keypoint, Eigen::Vector3d* ray) const {
    double rx = (keypoint(0) - c_.cu) / c_.fu;
    double ry = (keypoint(1) - c_.cv) / c_.fv;
    amo::undistort_normalized(c_, &rx, &ry);
    *ray = Eigen::Vector3d(rx, ry, 1.0);
    return true;
  }
  // aslam::Camera::getType / getParameters / getDistortion (pinhole: fu, fv, cu, cv)
  enum class Type { kPinhole = 0, kUnifiedProjection = 1 };
  Type getType() const { return Type::kPinhole; }
  Eigen::VectorXd getParameters() const {
    Eigen::VectorXd p(4);
    p(0) = c_.fu;
    p(1) = c_.fv;
    p(2) = c_.cu;
    p(3) = c_.cv;
    return p;
  }
  Distortion getDistortion() const {
    Eigen::VectorXd d(4);
    for (int k = 0; k < 4; ++k) d(k) = c_.dist[k];
    return Distortion(c_.distortion == 1   ? Distortion::Type::kRadTan
                      : c_.distortion == 2 ? Distortion::Type::kEquidistant
                                           : Distortion::Type::kNoDistortion,
                      d);
  }
  uint32_t imageWidth() const { return static_cast<uint32_t>(c_.width); }
  uint32_t imageHeight() const { return static_cast<uint32_t>(c_.height); }
  const amo_camera& parameters() const { return c_; }

 private:
  amo_camera c_;
};

class NCamera {
 public:
  typedef std::shared_ptr<NCamera> Ptr;
  NCamera(const amo_camera& camera, const Transformation& T_C_B)
      : camera_(new Camera(camera)), T_C_B_(T_C_B) {}
  const Camera& getCamera(size_t) const { return *camera_; }
  std::shared_ptr<const Camera> getCameraShared(size_t) const { return camera_; }
  const Transformation& get_T_C_B(size_t) const { return T_C_B_; }

 private:
  std::shared_ptr<Camera> camera_;
  Transformation T_C_B_;
};

}  // namespace aslam

#endif  // ORACLE_REFKIT_ASLAM_CAMERAS_CAMERA_H_
