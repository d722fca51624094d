/*
 * oracle/ref_loops_densify.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry point over the reference's OWN stereo::Densifier::computePointCloud
 * (aerial_mapper_dense_pcl/src/densifier.cpp, compiled unchanged from /root/reference; see
 * refkit/refkit.h).  Same arguments as the restated oracle's amo_densify (amo_ortho.cc).
 * Block matching (OpenCV's StereoBM / StereoSGBM behind block-matching-*.cpp) is out of scope:
 * the two wrappers' computeDisparityMap are defined empty here so that the densifier links.
 */
#include <aerial-mapper-dense-pcl/densifier.h>

#include "amo_types.h"
#include "refkit/refkit.h"

namespace stereo {
void BlockMatchingBM::computeDisparityMap(const RectifiedStereoPair&, DensifiedStereoPair*) const {}
void BlockMatchingSGBM::computeDisparityMap(const RectifiedStereoPair&, DensifiedStereoPair*) const {}
}  // namespace stereo

extern "C" {

/* K, R_G_C row-major 3x3; disparity float32 rows of disp_step BYTES; image_left 8UC1 rows of
 * img_step bytes; xyz_out / intensity_out sized for width*height points.  Returns the number
 * of points the reference pushed (point_cloud_eigen / point_cloud_intensities), -1 if one of
 * its CHECKs failed. */
long amr_densify(const float* disparity, size_t disp_step, const uint8_t* image_left,
                 size_t img_step, int width, int height, const double* K, double baseline,
                 const double* R_G_C, const double* t_G_C1, double* xyz_out,
                 int32_t* intensity_out) {
  refkit::check_reset();
  stereo::StereoRigParameters rig;
  stereo::RectifiedStereoPair rectified;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      rig.K(i, j) = K[3 * i + j];
      rectified.R_G_C(i, j) = R_G_C[3 * i + j];
    }
  rig.t_G_C1 = Eigen::Vector3d(t_G_C1[0], t_G_C1[1], t_G_C1[2]);
  rectified.baseline = baseline;
  rectified.image_left = cv::Mat(height, width, image_left, img_step);
  stereo::DensifiedStereoPair densified;
  densified.disparity_map =
      cv::Mat(height, width, reinterpret_cast<const uint8_t*>(disparity), disp_step);
  sensor_msgs::PointCloud2 cloud_ros;
  cloud_ros.point_step = 16;  // x, y, z, intensity
  // (the loop advances its offset BEFORE writing a point: one extra record)
  cloud_ros.data.assign((static_cast<size_t>(width) * height + 1) * cloud_ros.point_step, 0);
  const stereo::Densifier densifier(stereo::BlockMatchingParameters(), cv::Size(width, height));
  densifier.computePointCloud(rig, rectified, &densified, cloud_ros);
  if (refkit::check_state().failed) return -1;
  const size_t n = densified.point_cloud_eigen.size();
  if (densified.point_cloud_intensities.size() != n) return -1;
  for (size_t k = 0; k < n; ++k) {
    xyz_out[3 * k + 0] = densified.point_cloud_eigen[k](0);
    xyz_out[3 * k + 1] = densified.point_cloud_eigen[k](1);
    xyz_out[3 * k + 2] = densified.point_cloud_eigen[k](2);
    intensity_out[k] = densified.point_cloud_intensities[k];
  }
  return static_cast<long>(n);
}

}  // extern "C"
