// oracle/refkit: stand-in for aslam_cv2's mapped undistorter (see ../../refkit.h), which the
// forward mosaic runs every frame through.  As in the restated oracle (amo_forward.cc): the
// identity for a camera without distortion, otherwise a bilinear remap through the distortion
// model with the input intrinsics kept (amo::undistort_image; NOT pinned by this build, and a
// documented deviation from aslam's rescaled output camera).  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_ASLAM_UNDISTORTER_MAPPED_H_
#define ORACLE_REFKIT_ASLAM_UNDISTORTER_MAPPED_H_

#include <memory>

#include <aslam/cameras/camera.h>
#include <opencv2/highgui/highgui.hpp>

namespace aslam {

enum class InterpolationMethod { NearestNeighbor, Linear, Cubic, Lanczos };

class MappedUndistorter {
 public:
  explicit MappedUndistorter(const amo_camera& c) : c_(c) {}
  void processImage(const cv::Mat& input, cv::Mat* output) const {
    if (c_.distortion == AMO_DIST_NONE) {
      *output = input.clone();
      return;
    }
    const amo::Image8 in = {input.data, input.step, input.cols, input.rows, input.channels()};
    std::vector<uint8_t> out;
    amo::undistort_image(c_, in, &out);
    cv::Mat o(input.rows, input.cols, input.type());
    std::memcpy(o.data, out.data(), out.size());
    *output = o;
  }

 private:
  amo_camera c_;
};

inline std::unique_ptr<MappedUndistorter> createMappedUndistorter(const Camera& camera, float /*scale*/,
                                                                  float /*alpha*/,
                                                                  InterpolationMethod) {
  return std::unique_ptr<MappedUndistorter>(new MappedUndistorter(camera.parameters()));
}

}  // namespace aslam

#endif  // ORACLE_REFKIT_ASLAM_UNDISTORTER_MAPPED_H_
