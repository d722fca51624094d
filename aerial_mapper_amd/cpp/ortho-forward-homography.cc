// ortho::OrthoForwardHomography over the C ABI
// (see include/aerial-mapper-ortho/ortho-forward-homography.h).
#include "aerial-mapper-ortho/ortho-forward-homography.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "shim_common.h"

namespace ortho {

using amhip_shim::describe_camera;
using amhip_shim::pose_to7;

OrthoForwardHomography::OrthoForwardHomography(const std::shared_ptr<aslam::NCamera>& ncameras,
                                               const Settings& settings)
    : ncameras_(ncameras), settings_(settings), mosaic_(nullptr) {
  if (!ncameras_) amhip_shim::fatal("OrthoForwardHomography", "CHECK(ncameras_)");
  amhip_mosaic_desc desc;
  desc.width_mosaic_pixels = static_cast<int32_t>(settings_.width_mosaic_pixels);
  desc.height_mosaic_pixels = static_cast<int32_t>(settings_.height_mosaic_pixels);
  desc.ground_plane_elevation_m = settings_.ground_plane_elevation_m;
  for (int k = 0; k < 3; ++k) desc.origin[k] = settings_.origin(k);
  const amhip_camera cam = describe_camera(ncameras_->getCamera(kFrameIdx));
  const int device = amhip_shim::default_device();
  amhip_shim::check_status(amhip_mosaic_create(&desc, &cam, device, &mosaic_),
                           "OrthoForwardHomography");
  const size_t n = settings_.width_mosaic_pixels * settings_.height_mosaic_pixels;
  result_.assign(3 * n, 0);
  result_mask_.assign(n, 0);
}

OrthoForwardHomography::~OrthoForwardHomography() {
  if (mosaic_) amhip_mosaic_destroy(mosaic_);
}

void OrthoForwardHomography::updateOrthomosaic(const Pose& T_G_B, const Image& image) {
  double tgb[7], tcb[7], tgc[7];
  pose_to7(T_G_B, tgb);
  pose_to7(ncameras_->get_T_C_B(kFrameIdx), tcb);
  amhip_compose_T_G_C(tgb, tcb, 1, tgc);  // T_G_B * T_C_B^-1 (:83-84)
  amhip_shim::check_status(
      amhip_mosaic_update(mosaic_, tgc, image.data, static_cast<size_t>(image.step),
                          image.channels(), result_.data(), result_mask_.data()),
      "OrthoForwardHomography::updateOrthomosaic");
  writeOutput();  // cv::imwrite(settings_.filename_mosaic_output, result_) (:130)
}

void OrthoForwardHomography::batch(const Poses& T_G_Bs, const Images& images) {
  // the reference loops over images.size() and indexes T_G_Bs with it (:138-142)
  if (T_G_Bs.size() < images.size())
    amhip_shim::fatal("OrthoForwardHomography::batch", "fewer poses than images");
  const size_t F = images.size();
  std::vector<double> tgb(7 * F + 7), tgc(7 * F + 7);
  for (size_t f = 0; f < F; ++f) pose_to7(T_G_Bs[f], &tgb[7 * f]);
  double tcb[7];
  pose_to7(ncameras_->get_T_C_B(kFrameIdx), tcb);
  amhip_compose_T_G_C(tgb.data(), tcb, F, tgc.data());
  std::vector<const void*> data(F + 1);
  std::vector<size_t> steps(F + 1);
  int channels = 1;
  for (size_t f = 0; f < F; ++f) {
    data[f] = images[f].data;
    steps[f] = static_cast<size_t>(images[f].step);
    if (f == 0) channels = images[f].channels();
    if (images[f].channels() != channels)
      amhip_shim::fatal("OrthoForwardHomography::batch", "mixed gray / colour frames");
  }
  amhip_shim::check_status(
      amhip_mosaic_batch(mosaic_, tgc.data(), F, data.data(), steps.data(), channels,
                         result_.data(), result_mask_.data()),
      "OrthoForwardHomography::batch");
  writeOutput();  // cv::imwrite(settings_.filename_mosaic_output, result_) (:188)
}

cv::Mat OrthoForwardHomography::result8() const {
  const int h = static_cast<int>(settings_.height_mosaic_pixels);
  const int w = static_cast<int>(settings_.width_mosaic_pixels);
#if AERIAL_MAPPER_REAL_DEPS
  cv::Mat out(h, w, CV_8UC3);
#else
  cv::Mat out(h, w, 3);
#endif
  for (int y = 0; y < h; ++y) {
    uint8_t* row = out.data + static_cast<size_t>(y) * out.step;
    const int16_t* src = result_.data() + static_cast<size_t>(y) * w * 3;
    for (int k = 0; k < 3 * w; ++k) {
      const int v = src[k];
      row[k] = static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  return out;
}

void OrthoForwardHomography::writeOutput() const {
  if (settings_.filename_mosaic_output.empty()) return;
  // Without OpenCV there is no JPEG encoder: the 8-bit mosaic goes out as a
  // binary PPM next to the requested name.  A catkin build calls
  // cv::imwrite(settings_.filename_mosaic_output, result8()) here instead.
  const cv::Mat img = result8();
  const std::string name = settings_.filename_mosaic_output + ".ppm";
  std::FILE* f = std::fopen(name.c_str(), "wb");
  if (!f) return;
  std::fprintf(f, "P6\n%d %d\n255\n", img.cols, img.rows);
  for (int y = 0; y < img.rows; ++y)
    std::fwrite(img.data + static_cast<size_t>(y) * img.step, 1, static_cast<size_t>(img.cols) * 3, f);
  std::fclose(f);
}

}  // namespace ortho
