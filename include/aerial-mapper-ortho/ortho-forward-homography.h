// ortho::OrthoForwardHomography on MI355X -- drop-in for the reference class
// (aerial_mapper_ortho/include/aerial-mapper-ortho/ortho-forward-homography.h:33-86):
// same namespace, Settings and public signatures, so
// main-ortho-forward-homography.cc:80-102 compiles against it unchanged.
// What stays with the caller / the reference: the OpenCV windows
// (showOrthomosaicCvWindow), the ROS image publishers and cv::imwrite of the
// result -- this class keeps result_ (CV_16SC3) and result_mask_ (CV_8U) and
// offers accessors for them (an extension; the reference's are private).
// NOTE (as in the reference): two more headers define a struct named
// ortho::Settings; include only one of them per translation unit.
#ifndef AERIAL_MAPPER_HIP_ORTHO_FORWARD_HOMOGRAPHY_H_
#define AERIAL_MAPPER_HIP_ORTHO_FORWARD_HOMOGRAPHY_H_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "aerial-mapper-deps.h"
#include "aerial-mapper-io/aerial-mapper-io.h"
// The reference's header brings <ros/ros.h> and glog in (ortho-forward-homography.h:26, via the
// aslam headers), and main-ortho-forward-homography.cc:47,65 relies on that (ros::init, CHECK):
// a host that has them gets them here too.
#if defined(__has_include)
#if __has_include(<ros/ros.h>)
#include <ros/ros.h>
#endif
#if __has_include(<glog/logging.h>)
#include <glog/logging.h>
#endif
#endif

struct amhip_mosaic;

namespace ortho {

struct Settings {
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  bool batch = true;
  double ground_plane_elevation_m = 414.0;
  size_t width_mosaic_pixels = 1000;
  size_t height_mosaic_pixels = 1000;
  Eigen::Vector3d origin{0.0, 0.0, 0.0};
  std::string nframe_id = "map";
  std::string filename_mosaic_output = "/tmp/result.jpg";
};

class OrthoForwardHomography {
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  OrthoForwardHomography(const std::shared_ptr<aslam::NCamera>& ncameras,
                         const Settings& settings);
  ~OrthoForwardHomography();
  OrthoForwardHomography(const OrthoForwardHomography&) = delete;
  OrthoForwardHomography& operator=(const OrthoForwardHomography&) = delete;

  void updateOrthomosaic(const Pose& T_G_B, const Image& image);
  void batch(const Poses& T_G_Bs, const Images& images);

  // --- extensions -----------------------------------------------------------
  // result_: height x width x 3 int16 (CV_16SC3, row-major); result_mask_:
  // height x width uint8.  Valid after batch() / updateOrthomosaic().
  const std::vector<int16_t>& result() const { return result_; }
  const std::vector<uint8_t>& result_mask() const { return result_mask_; }
  // result_ converted like cv::Mat::convertTo(CV_8UC3) (saturating), for
  // imshow / imwrite / sensor_msgs::fillImage on the caller's side.
  cv::Mat result8() const;

 private:
  void writeOutput() const;

  static constexpr size_t kFrameIdx = 0u;
  std::shared_ptr<aslam::NCamera> ncameras_;
  Settings settings_;
  amhip_mosaic* mosaic_;
  std::vector<int16_t> result_;
  std::vector<uint8_t> result_mask_;
};

}  // namespace ortho
#endif  // AERIAL_MAPPER_HIP_ORTHO_FORWARD_HOMOGRAPHY_H_
