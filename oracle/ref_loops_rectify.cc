/*
 * oracle/ref_loops_rectify.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry point over the reference's OWN stereo::Rectifier::rectifyStereoPair
 * (aerial_mapper_dense_pcl/src/rectifier.cpp, compiled unchanged from /root/reference; see
 * refkit/refkit.h).  Same arguments as the restated oracle's amo_rectify_stereo_pair
 * (amo_rectify.cc).  Pins the FLOW of that file (which products, inverses, casts, per-pixel
 * operations, in which sequence); the Eigen / OpenCV arithmetic underneath is refkit's, i.e.
 * the oracle's adopted definitions (amo_rectify.h).
 */
#include <aerial-mapper-dense-pcl/rectifier.h>

#include "amo_types.h"
#include "refkit/refkit.h"

extern "C" {

int amr_rectify_stereo_pair(const double* K, const double* R1, const double* R2, const double* t1,
                            const double* t2, int width, int height, const uint8_t* left,
                            size_t left_step, const uint8_t* right, size_t right_step,
                            double* R_G_C_out, double* baseline_out, float* maps,
                            uint8_t* rect_left, uint8_t* rect_right, uint8_t* mask) {
  refkit::check_reset();
  stereo::StereoRigParameters rig;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      rig.K(i, j) = K[3 * i + j];
      rig.R_G_C1(i, j) = R1[3 * i + j];
      rig.R_G_C2(i, j) = R2[3 * i + j];
    }
  rig.t_G_C1 = Eigen::Vector3d(t1[0], t1[1], t1[2]);
  rig.t_G_C2 = Eigen::Vector3d(t2[0], t2[1], t2[2]);
  rig.image_size = cv::Size(width, height);
  stereo::Rectifier rectifier(cv::Size(width, height));
  stereo::RectifiedStereoPair out;
  rectifier.rectifyStereoPair(rig, cv::Mat(height, width, left, left_step),
                              cv::Mat(height, width, right, right_step), &out);
  const size_t n = static_cast<size_t>(width) * height;
  *baseline_out = out.baseline;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R_G_C_out[3 * i + j] = out.R_G_C(i, j);
  for (int v = 0; v < height; ++v)
    for (int u = 0; u < width; ++u) {
      const size_t o = static_cast<size_t>(v) * width + u;
      maps[o] = rectifier.map_rectify_1_x_.at<float>(v, u);
      maps[n + o] = rectifier.map_rectify_1_y_.at<float>(v, u);
      maps[2 * n + o] = rectifier.map_rectify_2_x_.at<float>(v, u);
      maps[3 * n + o] = rectifier.map_rectify_2_y_.at<float>(v, u);
      rect_left[o] = out.image_left.at<uchar>(v, u);
      rect_right[o] = out.image_right.at<uchar>(v, u);
      mask[o] = out.mask.at<uchar>(v, u);
    }
  return refkit::check_state().failed ? AMO_ERR_EXACT_HIT : AMO_OK;
}

}  // extern "C"
