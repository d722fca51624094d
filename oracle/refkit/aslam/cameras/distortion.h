// oracle/refkit: stand-in for <aslam/cameras/distortion.h> (see ../../refkit.h): the type tag
// and parameter vector aslam_cv2's Distortion base class exposes (aslam::Distortion::Type
// {kNoDistortion, kEquidistant, kFisheye, kRadTan}, getType(), getParameters()).
// TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_ASLAM_CAMERAS_DISTORTION_H_
#define ORACLE_REFKIT_ASLAM_CAMERAS_DISTORTION_H_
#include <Eigen/Core>
namespace aslam {
class Distortion {
 public:
  enum class Type { kNoDistortion = 0, kEquidistant = 1, kFisheye = 2, kRadTan = 3 };
  Distortion(Type type, const Eigen::VectorXd& params) : type_(type), params_(params) {}
  Type getType() const { return type_; }
  const Eigen::VectorXd& getParameters() const { return params_; }

 private:
  Type type_;
  Eigen::VectorXd params_;
};
}  // namespace aslam
#endif
