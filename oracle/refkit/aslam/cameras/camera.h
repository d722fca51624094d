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

#include "../../../amo_compat.h"

namespace kindr {
namespace minimal {

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
  const amo::Pose& pose() const { return p_; }

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
  uint32_t imageWidth() const { return static_cast<uint32_t>(c_.width); }
  uint32_t imageHeight() const { return static_cast<uint32_t>(c_.height); }

 private:
  amo_camera c_;
};

class NCamera {
 public:
  typedef std::shared_ptr<NCamera> Ptr;
  NCamera(const amo_camera& camera, const Transformation& T_C_B) : camera_(camera), T_C_B_(T_C_B) {}
  const Camera& getCamera(size_t) const { return camera_; }
  const Transformation& get_T_C_B(size_t) const { return T_C_B_; }

 private:
  Camera camera_;
  Transformation T_C_B_;
};

}  // namespace aslam

#endif  // ORACLE_REFKIT_ASLAM_CAMERAS_CAMERA_H_
